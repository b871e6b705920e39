#!/usr/bin/env python3
"""Register / scratch report of conv3d_wf.hip for gfx950 (no GPU needed: hipcc cross-compiles).

    python tools/wf_resource_report.py > profiles/r04_conv_wf_resource_usage.txt

Per kernel instantiation: the compiler's resource-usage remark (-Rpass-analysis=kernel-resource-usage) and, from the ISA
(-save-temps), WHERE the scratch (spill) instructions sit relative to the chunk loop -- the chunk loop is the first 216 (composed-LL mode: 96)
v_mfma instructions of a kernel (its body is unrolled over two chunks)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "tmdiff_amd", "csrc", "conv3d_wf.hip")
with tempfile.TemporaryDirectory() as tmp:
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{ROOT}/include", f"-I{ROOT}/tmdiff_amd/csrc",
           "-x", "hip", "-c", src, "-o", "wf.o", "-Rpass-analysis=kernel-resource-usage", "-save-temps"]
    p = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    rem = {}
    cur = None
    for line in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1); rem[cur] = {}
        m = re.search(r"remark: [^:]*:\d+:\d+:\s+(\w[\w \[\]/]*): (\S+)", line)
        if m and cur:
            rem[cur][m.group(1).strip()] = m.group(2)
    asm = open(os.path.join(tmp, "conv3d_wf-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
print("# conv3d_wf.hip, gfx950, hipcc -O3: resource usage per instantiation and position of the scratch (spill) instructions")
print("# template arguments: <TT (band tiles per workgroup), TH, TW, PAIR (two images per tile), LLM (composed Conv_0 + LL mode)>")
for i, l in enumerate(asm):
    m = re.match(r"^(_ZN12_GLOBAL__N_116conv3d_wf_kernel\S*):", l)
    if not m:
        continue
    name = m.group(1)
    end = next(j for j in range(i, len(asm)) if asm[j].strip().startswith(".Lfunc_end"))
    body = asm[i:end]
    targs = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)E", name).groups()
    mf = [k for k, x in enumerate(body) if "v_mfma" in x]
    sc = [k for k, x in enumerate(body) if re.search(r"\bscratch_(load|store)", x)]
    # the chunk loop's body is unrolled over two chunks: 2 x 54 K-steps x 2 sub-tiles = 216 MFMAs (composed-LL mode: 2 x 24 x 2 = 96);
    # further MFMAs belong to the folded residual convolution behind the loop (desc.rc_*: 8 blocks x 16 K-steps)
    nmain = 96 if targs[4] == "1" else 216
    lo, hi = mf[0], mf[nmain - 1]
    rc_hi = mf[-1]
    inside = [k for k in sc if lo <= k <= hi]
    in_rc = [k for k in sc if hi < k <= rc_hi]
    r = rem.get(name, {})
    print(f"\nconv3d_wf_kernel<{', '.join(targs[:3])}, {'true' if targs[3] == '1' else 'false'}, {'true' if targs[4] == '1' else 'false'}>")
    print("  " + "  ".join(f"{k}: {v}" for k, v in r.items() if k in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")))
    print(f"  ISA: {len(body)} lines, {len(mf)} v_mfma ({nmain} = the chunk loop, lines {lo}-{hi}" +
          (f"; {len(mf) - nmain} = the folded residual convolution behind it" if len(mf) > nmain else "") +
          f"), {len(sc)} scratch_load/store instructions over all epilogue variants: {sum(1 for k in sc if k < lo)} in the set-up before the "
          f"loop (once per workgroup), {len(inside)} INSIDE THE CHUNK LOOP, {len(in_rc)} in / between loop and residual-convolution phase, "
          f"{sum(1 for k in sc if k > rc_hi)} after the last MFMA (epilogue)")
