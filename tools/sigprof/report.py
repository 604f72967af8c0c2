"""Symbolise + aggregate a sigprof sample file (see sigprof.cpp). Usage: python tools/sigprof/report.py samples.txt [lib.so ...]"""
import bisect, collections, os, re, subprocess, sys
path, libs = sys.argv[1], sys.argv[2:]
tables = {}
for lib in libs:
    syms = []
    for line in subprocess.run(["nm", "-C", "--defined-only", "-n", lib], capture_output=True, text=True).stdout.splitlines():
        m = re.match(r"([0-9a-f]+) [tTwW] (.*)", line)
        if m:
            syms.append((int(m.group(1), 16), m.group(2)))
    tables[os.path.basename(lib)] = syms
def name(frame):
    m = re.match(r"(.*)\+0x([0-9a-f]+)\[(.*)\]", frame)
    if not m:
        return frame
    lib, off, dl = m.group(1), int(m.group(2), 16), m.group(3)
    if lib in tables and tables[lib]:
        i = bisect.bisect_right(tables[lib], (off, "￿")) - 1
        if i >= 0:
            s = tables[lib][i][1]
            s = re.sub(r"\(.*", "", s) if not s.startswith("operator") else s
            return lib.replace(".so", "") + ":" + s[:90]
    return lib.replace(".so", "").split(".")[0] + ":" + (dl if dl != "?" else hex(off))
self_c, incl_c, stacks, total = collections.Counter(), collections.Counter(), collections.Counter(), 0
for line in open(path):
    if line.startswith("#") or not line.strip():
        continue
    cnt, rest = line.split(" ", 1)
    cnt = int(cnt)
    frames = [name(f) for f in rest.strip().split(";")]
    total += cnt
    self_c[frames[0]] += cnt
    for f in set(frames):
        incl_c[f] += cnt
    stacks[" < ".join(frames[:5])] += cnt
print("%d samples" % total)
print("-- self (innermost frame)")
for k, v in self_c.most_common(22): print("%5.1f%%  %s" % (100.0 * v / total, k))
print("-- inclusive")
for k, v in incl_c.most_common(30): print("%5.1f%%  %s" % (100.0 * v / total, k))
print("-- stacks (innermost first)")
for k, v in stacks.most_common(18): print("%5.1f%%  %s" % (100.0 * v / total, k))
