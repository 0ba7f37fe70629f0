"""Are the device kernels of two source trees the same machine code?

    python tools/compare_isa.py <git-commit>            # that commit's csrc/ against the working tree

Compiles every translation unit of ttt-video-dit_amd/csrc to gfx950 assembly (device only) in both trees and compares
the instruction streams kernel by kernel (labels normalised, comments dropped).  Used to show that a refactor - e.g.
adding an opt-in template variant next to a kernel that has been validated on the hardware - leaves the default-dispatched
code bit-for-bit alone when no GPU is at hand to re-run the parity tests.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNITS = ["ttt_generic", "ttt_mfma", "ttt_mfma2", "ttt_mfma16", "ttt_mfma_bwd", "ttt_mfma_bwd2", "ttt_prepost", "attn_fwd", "attn_bwd", "attn_pre"]


def asm(src_dir, unit, out):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", unit + ".hip", "-o", out],
                   cwd=src_dir, stderr=subprocess.DEVNULL, check=True)
    kernels, cur = {}, None
    for line in open(out).read().split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
        elif cur is not None:
            if "s_endpgm" in line:
                cur = None
                continue
            t = re.sub(r";.*", "", line).strip()
            if t and not t.startswith("."):
                kernels[cur].append(re.sub(r"\.LBB\d+_", ".LBB_", t))
    return kernels


def main():
    commit = sys.argv[1]
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(f"git -C {ROOT} archive {commit} ttt-video-dit_amd/csrc include | tar -x -C {tmp}", shell=True, check=True)
        old_dir, new_dir = os.path.join(tmp, "ttt-video-dit_amd", "csrc"), os.path.join(ROOT, "ttt-video-dit_amd", "csrc")
        changed = 0
        for unit in UNITS:
            if not os.path.exists(os.path.join(old_dir, unit + ".hip")):
                print(f"{unit}: not in {commit}")
                continue
            a, b = asm(old_dir, unit, os.path.join(tmp, "a.s")), asm(new_dir, unit, os.path.join(tmp, "b.s"))
            by_code = {tuple(v): k for k, v in b.items()}
            for k, v in a.items():
                if k in b and b[k] == v:
                    continue
                if tuple(v) in by_code:
                    print(f"{unit}: {k} identical, now named {by_code[tuple(v)]}")
                else:
                    print(f"{unit}: {k} CHANGED" if k in b else f"{unit}: {k} removed / changed under another name")
                    changed += 1
            print(f"{unit}: {len(a)} kernels compared, {max(len(b) - len(a), 0)} new")
        print("no kernel of the old tree changed" if not changed else f"{changed} kernel(s) changed")
        return 1 if changed else 0


if __name__ == "__main__":
    sys.exit(main())
