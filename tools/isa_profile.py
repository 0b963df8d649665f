"""Static instruction profile of one kernel of an assembly listing compiled with -gline-tables-only:
VALU / SALU / VMEM / LDS instruction counts per source line range.  usage:
  hipcc ... -gline-tables-only --cuda-device-only -S -o k.s srt_kernels.hip
  python tools/isa_profile.py k.s _Z17srt_render_kernelILb0ELb0ELb1EEv10RenderArgs [bucket_edges...]"""
import collections, re, sys

path, kernel = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2))
cur = 0
per = collections.defaultdict(lambda: collections.Counter())
for l in lines[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        f = files.get(int(m.group(1)), "?")
        cur = int(m.group(2)) if "srt_kernels" in f else 100000 + int(m.group(1))  # other files: one bucket per file
        continue
    t = l.strip()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    kind = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "vmem" if op.startswith(("buffer_", "global_", "scratch_", "flat_"))
            else "lds" if op.startswith("ds_") else "other")
    per[cur][kind] += 1
    if op.endswith("_f64") or "_f64_" in op or op.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64")):
        per[cur]["slow"] += 1
edges = [int(x) for x in sys.argv[3:]] or None

tot = collections.Counter()
for ln in sorted(per):
    tot.update(per[ln])
print("total", dict(tot))
if edges:
    b = collections.defaultdict(collections.Counter)
    for ln in per:
        k = ln if ln >= 100000 else max([e for e in edges if e <= ln], default=0)
        b[k].update(per[ln])
    for k in sorted(b):
        print("from line %6d %s: %s" % (k, files.get(k - 100000, "") if k >= 100000 else "", dict(b[k])))
else:
    for ln in sorted(per):
        print(ln, dict(per[ln]))
