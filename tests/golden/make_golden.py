#!/usr/bin/env python3
"""Collects the golden vectors for the pattern-matching hot path into tests/golden/.

Run once in the build container (where /root/reference is mounted); the GPU box has no
reference tree, so everything the tests need is committed here as DATA:

* the reference's own fixture/data files (inputs and expected outputs its tests hold:
  tests/fixtures/**, tests/data/*, example-minimal/*, the small example-workflow goldens);
* a reduced copy of the 2 x 4 MB example-workflow FASTQs: the 24 record pairs the golden
  output holds plus every 40th other pair, in file order.  Because the subset keeps every
  record named in the golden JSON and preserves order, the golden's 36-row hit list, its
  per-pattern counts and its extracted FASTQs remain the expected output for the subset;
  only `number_of_records_searched` / `number_of_characters_searched` change, and those
  are recomputed from the subset itself (they are plain sums of input sizes).

No reference SOURCE is copied: only data files.
"""
import gzip
import json
import os
import shutil
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def cp(rel, dst_rel=None):
    dst = os.path.join(HERE, dst_rel or rel)
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    shutil.copyfile(os.path.join(REF, rel), dst)


def read_fastq(path):
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    assert len(lines) % 4 == 0
    return [lines[i:i + 4] for i in range(0, len(lines), 4)]


def main():
    for d in ("tests/fixtures/input", "tests/fixtures/extract", "tests/fixtures/tag", "tests/data"):
        for fn in sorted(os.listdir(os.path.join(REF, d))):
            cp(os.path.join(d, fn), os.path.join(d.replace("tests/", "", 1), fn))
    for fn in ("kmers.txt", "sample.fasta", "sample.sam"):
        cp(os.path.join("example-minimal", fn))
    wf = "example-workflow"
    cp(f"{wf}/significant_kmers.txt")
    cp(f"{wf}/logs/mutant_extracted.stats.json")
    for fn in ("mutant_extracted_1.fastq", "mutant_extracted_2.fastq",
               "mutant_extracted.sorted.sam", "mutant_extracted.sorted.tagged.sam"):
        cp(f"{wf}/output/{fn}")
    # reduced workflow FASTQs
    r1 = read_fastq(os.path.join(REF, wf, "data/mutant_R1.fastq"))
    r2 = read_fastq(os.path.join(REF, wf, "data/mutant_R2.fastq"))
    assert len(r1) == len(r2) == 12480
    gold = json.load(open(os.path.join(REF, wf, "logs/mutant_extracted.stats.json")))
    hit_ids = {m["record_id"].encode() for m in gold["matching_records"]}
    ext = {rec[0][1:] for rec in read_fastq(os.path.join(REF, wf, "output/mutant_extracted_1.fastq"))}
    keep = [i for i in range(len(r1))
            if r1[i][0][1:] in hit_ids or r2[i][0][1:] in hit_ids or r1[i][0][1:] in ext or i % 40 == 0]
    for name, recs in (("mutant_R1.subset.fastq.gz", r1), ("mutant_R2.subset.fastq.gz", r2)):
        out = os.path.join(HERE, wf, "data", name)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        with open(out, "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, filename="") as g:
            for i in keep:
                g.write(b"\n".join(recs[i]) + b"\n")
    meta = {"pairs_total_in_reference": len(r1), "pairs_kept": len(keep),
            "bases_kept": sum(len(r1[i][1]) + len(r2[i][1]) for i in keep)}
    json.dump(meta, open(os.path.join(HERE, wf, "data", "subset.meta.json"), "w"), indent=1)
    print(meta)


if __name__ == "__main__":
    main()
