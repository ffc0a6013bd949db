import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from merkurio_amd import native as mk
pats = mk.parse_pattern_list(kmer_seq=[b"ACGTTGCA", b"TTTTGGGG"])
m = mk.Matcher(pats)
text = b">a x\nACGTTGCATT\nTTGGGGAC\n>b\nAAAAAAAA\n>c\n\n>d\nACGTT\nGCA\n"
r = m.extract_window([{"text": text}], fmt=mk.MK_TEXT_FASTA, want=("kept", "tail"))
print(r)
r = m.extract_window([{"text": text}], fmt=mk.MK_TEXT_FASTA, logging=False)
print(r["keep"], r["counters"])
seqs=[b"ACGTTGCATTTTGGGGAC", b"AAAAAAAA", b"", b"ACGTTGCA"]
print(m.extract_single(seqs))
