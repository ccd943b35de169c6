"""Build the HIP library in-tree: myzkp_amd/libmzk_hip.so (gfx950 only).

    python -m myzkp_amd.build [--force] [--tuning]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with gpurun.
--tuning builds myzkp_amd/libmzk_hip_tuning.so instead (-DMZK_TUNING): the A/B switches of tools/timing (MZK_* environment
variables) and the superseded kernels only they can reach exist there and nowhere else; load it with MZK_HIP_LIB=<that file>."""
import os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmzk_hip.so")
OUT_TUNING = os.path.join(HERE, "libmzk_hip_tuning.so")
SOURCES = ["mzk_api.hip", "mzk_multi.hip", "mzk_io.hip", "mzk_poly.hip", "mzk_ntt.hip", "mzk_msm.hip", "mzk_msm_row.hip", "mzk_kzg.hip", "mzk_merkle.hip", "mzk_g2.hip", "mzk_selftest.hip"]
TUNING_ONLY_SOURCES = ["mzk_msm_tail.hip"]      # the DPP-quad tails of round 2 (MZK_ROW_TAILS=0)
import glob
# every header of csrc/ plus the ABI header: a hand-kept list went stale once (mzk_glv.h)
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "mzk.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-gpu-rdc"]


def _newer(src, dst):
    return not os.path.exists(dst) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force=False, verbose=False, tuning=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    gen = os.path.join(ROOT, "tools", "gen_constants.py")
    consts = os.path.join(CSRC, "mzk_constants.h")
    if os.path.exists(gen) and _newer(gen, consts):
        subprocess.check_call([sys.executable, gen, consts])
    gen_asm = os.path.join(ROOT, "tools", "gen_field_asm.py")
    asm_h = os.path.join(CSRC, "mzk_field_asm.h")
    if os.path.exists(gen_asm) and (_newer(gen_asm, asm_h) or _newer(gen, asm_h)):
        subprocess.check_call([sys.executable, gen_asm, asm_h])
    gen_k = os.path.join(ROOT, "tools", "gen_keccak_asm.py")
    kec_h = os.path.join(CSRC, "mzk_keccak_asm.h")
    if os.path.exists(gen_k) and _newer(gen_k, kec_h):
        subprocess.check_call([sys.executable, gen_k, kec_h])
    objdir = os.path.join(HERE, "build_tuning" if tuning else "build")
    os.makedirs(objdir, exist_ok=True)
    out = OUT_TUNING if tuning else OUT
    flags = FLAGS + (["-DMZK_TUNING"] if tuning else [])
    objs, procs = [], []
    for s in SOURCES + (TUNING_ONLY_SOURCES if tuning else []):
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(src, obj) or any(_newer(h, obj) for h in hdrs):
            cmd = [hipcc] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + s)
    if procs or not os.path.exists(out):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, tuning="--tuning" in sys.argv))
