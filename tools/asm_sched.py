"""Print the instruction-class schedule of the hottest basic block of selected kernels (dev tool)."""
import re, subprocess, sys, os, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = tempfile.mkdtemp()
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-Wno-unused-value",
                "-save-temps", "-c", os.path.join(root, "protnote_amd/csrc/protnote_hip.hip"), "-o", "x.o"], cwd=d,
               stderr=subprocess.DEVNULL, check=True)
s = open(os.path.join(d, "protnote_hip-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
pats = sys.argv[1:] or ["gemm_nt_kernelILi1ELi0ELi2ELi2ELi2ELi2ELi32", "gemm_tn_kernelILi1ELi2E"]
for name in re.findall(r"^(_ZN2pn\w+):", s, flags=re.M):
    if not any(p in name for p in pats):
        continue
    i = s.index(name + ":"); j = s.index(".Lfunc_end", i); body = s[i:j]
    blocks = re.split(r"\n(\.LBB\d+_\d+):", body)
    cands = [(blocks[k + 1].count("v_mfma"), blocks[k], blocks[k + 1]) for k in range(1, len(blocks), 2)]
    cands = [c for c in cands if c[0] >= 32 and ("global_load" in c[2] or "buffer_load" in c[2])] or [max(cands)]
    best = cands[0]
    seq = []
    for l in best[2].split("\n"):
        l = l.strip()
        if not l or l[0] in ";.":
            continue
        op = l.split()[0]
        if op.startswith("v_mfma"): op = "M"
        elif op.startswith("global_load") or op.startswith("buffer_load"): op = "G"
        elif op.startswith("ds_read"): op = "r"
        elif op.startswith("ds_write"): op = "W"
        elif op == "s_waitcnt": op = "[" + l.split(None, 1)[1].split(";")[0].strip() + "]"
        elif op == "s_barrier": op = "|BAR|"
        else: op = "."
        seq.append(op)
    meta = s[j:j + 4000]
    print(name, "mfma:", best[0], best[1])
    print("".join(seq))
    k = s.find(".name:           " + name)
    m = re.search(r"\.vgpr_count:\s+(\d+)", s[k:k + 3000]) if k > 0 else None
    a = re.search(r"\.agpr_count:\s+(\d+)", s[k:k + 3000]) if k > 0 else None
    print("vgpr", m.group(1) if m else "?", "agpr", a.group(1) if a else "?")
