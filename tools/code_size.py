#!/usr/bin/env python3
"""tools/code_size.py [build dir] -- device code bytes per translation unit and kernel template (from the gfx950 code objects inside the .o files):
what the library's size is made of.  Round 6: the pruning decisions of DESIGN section 7 are read off this table."""
import collections, os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
bdir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "robotdynamics.jl_amd", "build")
tot = collections.Counter(); cnt = collections.Counter(); per_unit = {}
with tempfile.TemporaryDirectory() as td:
    for o in sorted(os.listdir(bdir)):
        if not o.endswith(".o"): continue
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", os.path.join(bdir, o), fat], check=True)
        if not os.path.exists(fat) or os.path.getsize(fat) == 0: continue
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True)
        if r.returncode: continue
        syms = subprocess.run([f"{LLVM}/llvm-readelf", "-sW", co], capture_output=True, text=True).stdout
        names, sizes = [], []
        for l in syms.splitlines():
            f = l.split()
            if len(f) >= 8 and f[3] == "FUNC": sizes.append(int(f[2])); names.append(f[7])
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        u = collections.Counter(); uc = collections.Counter()
        for s, n in zip(sizes, dem):
            n = re.sub(r"^void ", "", n); t = n.split("<")[0].split("(")[0]
            dt = "f32" if re.search(r"\bfloat\b", n) else "f64"
            u[(t, dt)] += s; uc[(t, dt)] += 1
        per_unit[o] = (os.path.getsize(co), u, uc)
        for k, v in u.items(): tot[k] += v; cnt[k] += uc[k]
        os.remove(fat); os.remove(co)
print(f"{'unit':20s} {'code object':>12s}")
for o, (sz, u, uc) in per_unit.items(): print(f"{o:20s} {sz:12d}  kernels {sum(uc.values())}")
print(f"{'TOTAL':20s} {sum(v[0] for v in per_unit.values()):12d}  kernels {sum(cnt.values())}\n")
print(f"{'template':38s} {'dtype':5s} {'kernels':>8s} {'bytes':>12s}")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]): print(f"{k[0]:38s} {k[1]:5s} {cnt[k]:8d} {v:12d}")
if len(sys.argv) > 2:
    o = sys.argv[2]; sz, u, uc = per_unit[o]
    print(f"\n{o}:")
    for k, v in sorted(u.items(), key=lambda kv: -kv[1]): print(f"{k[0]:38s} {k[1]:5s} {uc[k]:8d} {v:12d}")
