import sys, zlib
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from merkurio_amd import native as mk
from test_codec_cpu import corpora, raw_deflate
from textio import bam_like
c = mk.Codec()
for name, data in (("bam", bam_like(40000)), ("fastq text", corpora()["fastq"] * 8), ("skew", corpora()["skew"] * 4), ("ramp", corpora()["ramp"])):
    z1 = sum(len(raw_deflate(data[i:i + 65280], level=1)) + 26 for i in range(0, len(data), 65280))
    z6 = sum(len(raw_deflate(data[i:i + 65280], level=6)) + 26 for i in range(0, len(data), 65280))
    print(f"   {name}: device {len(data)/len(c.deflate(data)):.2f}  zlib-1 {len(data)/z1:.2f}  zlib-6 {len(data)/z6:.2f}")
