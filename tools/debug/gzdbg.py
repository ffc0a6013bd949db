import sys, zlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import _fastq_binned
from merkurio_amd import native as mk
data = _fastq_binned(100000)
c = mk.Codec()
for lvl in (1, 6):
    co = zlib.compressobj(lvl, zlib.DEFLATED, 31); gz = co.compress(data) + co.flush()
    t = c.gunzip(gz)
    print(lvl, c.gzip_info, t == data)
