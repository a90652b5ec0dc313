#!/usr/bin/env python3
"""Resolves the raw addresses of a glog crash report (rocprofv3's failure handler prints 'PC: @ 0x...' and '@ 0x...' frames without symbols)
against a saved /proc/<pid>/maps: library, offset, and the nearest exported / local symbol below it (nm -D / nm, llvm-symbolizer where it
knows more).   python3 tools/pmc_abort_symbolise.py run.log maps.txt"""
import bisect, os, re, subprocess, sys
log, maps = open(sys.argv[1]).read(), open(sys.argv[2]).read().split("\n")
segs = []
for line in maps:
    m = re.match(r"([0-9a-f]+)-([0-9a-f]+) (\S+) ([0-9a-f]+) \S+ \S+\s*(.*)", line)
    if m:
        segs.append((int(m.group(1), 16), int(m.group(2), 16), m.group(3), int(m.group(4), 16), m.group(5)))
print("\n".join(l for l in log.split("\n") if "Aborted" in l or "SIG" in l or "exit code" in l or l.startswith("pass ")))
syms = {}
def table(path):
    if path not in syms:
        out = []
        for cmd in (["nm", "-D", "--defined-only", path], ["nm", "--defined-only", path]):
            try:
                for l in subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode().split("\n"):
                    f = l.split()
                    if len(f) >= 3 and f[1] in "TtWwiI":
                        out.append((int(f[0], 16), f[2]))
            except OSError:
                pass
        syms[path] = sorted(set(out))
    return syms[path]
sym = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"
fault = re.search(r"SIG\w+ \(@(0x[0-9a-f]+)\)", log)
addrs = ([("faulting address", int(fault.group(1), 16))] if fault else []) + [("frame", int(a, 16)) for a in re.findall(r"@\s+(0x[0-9a-f]+)", log)]
for kind, a in addrs:
    hit = [s for s in segs if s[0] <= a < s[1]]
    if not hit:
        near = [s for s in segs if s[1] <= a]
        print(f"{kind} {a:#x}: not inside any mapping of the snapshot" + (f" (first byte behind {near[-1][4] or 'an anonymous mapping'} {near[-1][0]:#x}-{near[-1][1]:#x} {near[-1][2]})" if near and a - near[-1][1] < (1 << 21) else ""))
        continue
    lo, hi, perm, off, path = hit[0]
    if not path or not os.path.exists(path):
        print(f"{kind} {a:#x}: {path or 'anonymous mapping'} {lo:#x}-{hi:#x} {perm} (+{a - lo:#x})")
        continue
    base = min(s[0] - s[3] for s in segs if s[4] == path)
    rel = a - base
    t = table(path)
    i = bisect.bisect_right([x for x, _ in t], rel) - 1
    name = f"{t[i][1]}+{rel - t[i][0]:#x}" if i >= 0 else "?"
    try:
        dem = subprocess.run([sym, "-e", path, "-f", "-C", hex(rel)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=60).stdout.decode().split("\n")[0]
    except (OSError, subprocess.TimeoutExpired):
        dem = ""
    try:
        name = subprocess.run(["c++filt", name.split("+")[0]], stdout=subprocess.PIPE).stdout.decode().strip() + "+" + name.split("+")[1]
    except (OSError, IndexError):
        pass
    print(f"{kind} {a:#x}: {os.path.basename(path)} +{rel:#x}  {name}" + (f"  [{dem}]" if dem and dem != "??" else ""))
