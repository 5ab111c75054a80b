"""Build liblpcnet_hip.so (HIP kernels for gfx950 + C host shell) in-tree with hipcc/gcc.

    python -m lpcnet_amd.build [--force]

hipcc cross-compiles gfx950 without a GPU.  The flags matter for bit-exactness:
  -ffp-contract=off     every multiply and add is rounded separately (the reference's
                        reproducible generic-C flavour does the same)
  -fno-slp-vectorize    keep scalar f32 ops so the DPP quad broadcasts fold into v_mul_f32_dpp
                        (packed v_pk_mul_f32 would need the weights duplicated in register pairs)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblpcnet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

HIP_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
             "-fvisibility=hidden", "-I" + CSRC, "-I" + os.path.join(HERE, "..", "include")]
C_FLAGS = ["-O2", "-fPIC", "-ffp-contract=off", "-std=gnu11", "-Wall", "-fvisibility=hidden", "-DLPCNET_BUILD",
           "-I" + CSRC, "-I" + os.path.join(HERE, "..", "include")]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, prof=False):
    """prof=True builds liblpcnet_hip_prof.so with the in-kernel phase profiler compiled in
    (tools only; select it with LPCNET_HIP_LIB=<path>)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    srcs += [os.path.join(HERE, "..", "include", f) for f in ("lpcnet.h", "lpcnet_batch.h")]
    lib = LIB.replace(".so", "_prof.so") if prof else LIB
    if not force and not _newer(lib, srcs):
        return lib
    objdir = os.path.join(HERE, "build_prof" if prof else "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    # engine.hip + the sample kernel's variants, one translation unit per streams-per-workgroup value: built in parallel
    pf = ["-DLPCN_ENABLE_PROF=1"] + (["-DLPCN_PROF_MASK=" + os.environ["LPCN_PROF_MASK"]] if "LPCN_PROF_MASK" in os.environ else []) if prof else []
    extra = os.environ.get("LPCN_EXTRA_FLAGS", "").split()
    jobs = [([HIPCC] + HIP_FLAGS + pf + extra + ["-c", os.path.join(CSRC, "engine.hip"), "-o", os.path.join(objdir, "engine.o")])]
    for sv in (1, 2, 4):
        jobs.append([HIPCC] + HIP_FLAGS + pf + extra + [f"-DLPCN_S={sv}", "-c", os.path.join(CSRC, "sample_variants.hip"), "-o", os.path.join(objdir, f"sample_s{sv}.o")])
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs += [j[-1] for j in jobs]
    for c in ("api.c", "model_pack.c"):
        o = os.path.join(objdir, c[:-2] + ".o")
        run(["gcc"] + C_FLAGS + ["-c", os.path.join(CSRC, c), "-o", o])
        objs.append(o)
    run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs + ["-lpthread", "-lm"])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, prof="--prof" in sys.argv))
