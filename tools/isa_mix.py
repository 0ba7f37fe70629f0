"""Static instruction mix of the gfx950 kernels, loop by loop - a profiler substitute for when no GPU is at hand.

    python tools/isa_mix.py ttt_mfma_bwd2 [kernel-name-substring]      # a translation unit of ttt-video-dit_amd/csrc

Compiles the unit to gfx950 assembly (device only), splits each kernel into basic blocks, finds the natural loops
(backward branches) and prints per loop body, for ONE wave: instruction counts by class (MFMA, other VALU, transcendental,
LDS, VMEM / buffer, scalar, waitcnt, barriers) and a lower bound on the issue time of the SIMD,

    cycles >= waves_per_simd * sum(issue cycles of the wave's instructions)   and   >= MFMA pipe cycles,

with the issue costs of MI355X_MICROARCH.md (wave64 VALU 4 cycles, a transcendental ~5/3 of that, 32x32x16 bf16 MFMA 8 passes
x 4 = 32 cycles of the matrix pipe - 16 for the 16x16x32 shape -, LDS / VMEM 4 to issue).  It knows nothing about latencies
or dependencies: it bounds what scheduling can reach and shows whether a loop is MFMA-, VALU- or wait-dominated.

CAVEAT: a loop's span contains blocks that the steady state may jump over (a ragged-tail mask behind a uniform branch, a
checkpoint store every G-th step).  `--blocks` prints the innermost large loop basic block by basic block with the branch
that ends each block, so that skipped blocks can be told from the hot path (attention forward, revision 1: the 96-VALU
mask block is skipped by `s_cbranch_vccnz`; its dQ kernel has the mask if-converted INTO the hot block: 243 VALU).
"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ttt-video-dit_amd", "csrc")

TRANS = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "acc_mov"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op == "s_waitcnt":
        return "waitcnt"
    if op == "s_barrier":
        return "barrier"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_nop") or op.startswith("s_sleep"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def mfma_cycles(op):
    # passes x 4 cycles; 32x32x16 (bf16/f16) = 8 passes, 16x16x32 = 4 passes on gfx950; older K shapes listed for completeness
    if "32x32x16" in op:
        return 32
    if "16x16x32" in op:
        return 16
    if "32x32x8" in op:
        return 64 if "f32_32x32x8" in op and "bf16" not in op and "f16" not in op else 32
    if "16x16x16" in op:
        return 16
    if "32x32x2" in op or "32x32x1" in op:
        return 64
    if "16x16x4" in op or "16x16x1" in op:
        return 32
    if "4x4" in op:
        return 8
    return 32


ISSUE = {"valu": 4, "acc_mov": 4, "trans": 7, "lds": 4, "vmem": 4, "salu": 1, "waitcnt": 1, "barrier": 1, "branch": 1, "nop": 1, "other": 1, "mfma": 4}


def parse(asm_path):
    kernels, cur = {}, None
    for line in open(asm_path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = kernels[m.group(1)] = []
            continue
        if cur is None:
            continue
        t = re.sub(r";.*", "", line).strip()
        if not t:
            continue
        if t.startswith(".LBB") and t.endswith(":"):
            cur.append(("label", t[:-1]))
            continue
        if t.startswith("."):
            continue
        op = t.split()[0]
        cur.append(("inst", op, t))
        if op == "s_endpgm":
            cur = None
    return kernels


def loops(insts):
    """(start_index, end_index, label) for every backward branch (natural loop bodies, innermost have the shortest span)."""
    pos = {x[1]: i for i, x in enumerate(insts) if x[0] == "label"}
    out = []
    for i, x in enumerate(insts):
        if x[0] == "inst" and (x[1].startswith("s_cbranch") or x[1] == "s_branch"):
            tgt = x[2].split()[-1]
            if tgt in pos and pos[tgt] < i:
                out.append((pos[tgt], i, tgt))
    return out


def mix(insts):
    c, mfma_pipe, issue = Counter(), 0, 0
    for x in insts:
        if x[0] != "inst":
            continue
        k = classify(x[1])
        c[k] += 1
        issue += ISSUE[k]
        if k == "mfma":
            mfma_pipe += mfma_cycles(x[1])
    return c, mfma_pipe, issue


def blocks(insts, a, b, lab):
    """basic blocks of insts[a..b]: (label, instructions, terminating instruction text)"""
    out, cur = [], [lab]
    for x in insts[a:b + 1]:
        if x[0] == "label":
            if len(cur) > 1:
                out.append(cur)
            cur = [x[1]]
        else:
            cur.append(x)
            if x[1].startswith("s_cbranch") or x[1] == "s_branch":
                out.append(cur)
                cur = ["(fallthrough)"]
    if len(cur) > 1:
        out.append(cur)
    return out


def demangle(name):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        return name


def main():
    args = [a for a in sys.argv[1:] if a != "--blocks"]
    show_blocks = "--blocks" in sys.argv
    unit = args[0]
    pat = args[1] if len(args) > 1 else ""
    waves_per_simd = int(os.environ.get("WAVES_PER_SIMD", "2"))
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, unit + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", unit + ".hip", "-o", out],
                       cwd=CSRC, stderr=subprocess.DEVNULL, check=True)
        kernels = parse(out)
    for name, insts in kernels.items():
        pretty = demangle(name)
        if pat and pat not in pretty and pat not in name:
            continue
        c, pipe, issue = mix(insts)
        print(f"\n== {pretty[:150]}\n   whole kernel: {sum(c.values())} instructions  {dict(c)}")
        ls = sorted(loops(insts), key=lambda l: l[1] - l[0], reverse=True)
        for a, b, lab in ls:
            c, pipe, issue = mix(insts[a:b + 1])
            n = sum(c.values())
            if n < 40:
                continue
            print(f"   loop {lab:>10} [{n:5d} inst]  mfma {c['mfma']:4d} (pipe {pipe:6d} cyc)  valu {c['valu']:5d}  trans {c['trans']:4d}  acc_mov {c['acc_mov']:4d}  "
                  f"lds {c['lds']:4d}  vmem {c['vmem']:4d}  salu {c['salu']:4d}  wait {c['waitcnt']:4d}  barrier {c['barrier']:3d}  "
                  f"| issue/wave {issue:6d} cyc -> SIMD bound at {waves_per_simd} waves: max({waves_per_simd * issue}, {waves_per_simd * pipe}) cyc")
        big = [l for l in ls if l[1] - l[0] > 100]
        if show_blocks and big:
            a, b, lab = big[0] if os.environ.get("ISA_LOOP", "outer") == "outer" else big[-1]
            print(f"   basic blocks of loop {lab}:")
            for bl in blocks(insts, a, b, lab):
                c, pipe, issue = mix(bl[1:])
                scr = sum(1 for x in bl[1:] if x[1].startswith("scratch"))
                print(f"      {bl[0]:>14} n={len(bl) - 1:4d} valu={c['valu']:4d} trans={c['trans']:3d} mfma={c['mfma']:3d} lds={c['lds']:3d} "
                      f"vmem={c['vmem']:3d} (scratch {scr:2d}) wait={c['waitcnt']:3d}   ends: {bl[-1][2]}")


if __name__ == "__main__":
    main()
