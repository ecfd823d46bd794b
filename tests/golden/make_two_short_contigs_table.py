"""Extracts the 66-row expected record table of testdata/two_short_contigs.ctx from the
reference's own known-answer test (public/java/tests/.../utils/kmer/CortexGraphTest.java:71-136)
into tests/golden/two_short_contigs.expected.txt ("KMER cov0 cov1 edges0 edges1" per line).
Only data (inputs / expected outputs) is extracted.  Run in the build container, where
/root/reference exists; the GPU box only sees the committed .txt."""
import re
import sys

src = "/root/reference/public/java/tests/uk/ac/ox/well/cortexjdk/utils/kmer/CortexGraphTest.java"
pat = re.compile(r'SimpleCortexRecord\("([ACGT]+)",\s*new int\[\]\s*\{\s*(\d+),\s*(\d+)\s*\},\s*new String\[\]\s*\{"([^"]+)",\s*"([^"]+)"\}')
rows = [m.groups() for m in pat.finditer(open(src).read())]
assert len(rows) == 66, len(rows)
out = sys.argv[1] if len(sys.argv) > 1 else "tests/golden/two_short_contigs.expected.txt"
with open(out, "w") as f:
    for k, c0, c1, e0, e1 in rows:
        f.write(f"{k} {c0} {c1} {e0} {e1}\n")
print("wrote", len(rows), "rows to", out)
