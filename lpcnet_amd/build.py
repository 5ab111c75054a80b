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


def source_hashes(extra_flags=()):
    """(src, dev): sha1 prefixes over every source the library is built from (lpcnet_amd/csrc/*, the public headers, the
    compile flags) and over the DEVICE sources alone (*.hip, *.hip.h, *.inc and the headers they include -- everything in
    csrc/ except the host C files).  Both are baked into the library (lpcnet_hip_build_info()); `dev` ties a rocprofv3
    measurement under profiles/ to the kernels it was taken with (bench.py: kernel_source_hash), so a change to host code
    alone does not orphan the profiles."""
    import hashlib
    hs, hd = hashlib.sha1(), hashlib.sha1()
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith((".h", ".hip", ".inc", ".c")):
            continue
        data = f.encode() + open(os.path.join(CSRC, f), "rb").read()
        hs.update(data)
        if not f.endswith(".c"):
            hd.update(data)
    for f in ("lpcnet.h", "lpcnet_batch.h", "lpcnet_hip_state.h"):
        hs.update(f.encode() + open(os.path.join(HERE, "..", "include", f), "rb").read())
    flags = " ".join(x for x in HIP_FLAGS if not x.startswith("-I")) + " | " + " ".join(extra_flags)
    hs.update((flags + " | " + " ".join(x for x in C_FLAGS if not x.startswith("-I"))).encode())
    hd.update(flags.encode())
    return hs.hexdigest()[:16], hd.hexdigest()[:16]


def baked_hashes(lib):
    """The (src, dev) hashes a built library carries, read from the file without loading it (None if absent / older build)."""
    try:
        data = open(lib, "rb").read()
    except OSError:
        return None
    i = data.find(b"LPCN_BUILD_INFO: src=")
    if i < 0:
        return None
    txt = data[i:i + 96].split(b"\0", 1)[0].decode(errors="replace")
    try:
        kv = dict(x.split("=", 1) for x in txt.split()[1:])
        return kv["src"], kv["dev"]
    except Exception:
        return None


def build(force=False, verbose=True, prof=False):
    """prof=True builds liblpcnet_hip_prof.so with the in-kernel phase profiler compiled in
    (tools only; select it with LPCNET_HIP_LIB=<path>).  A library is up to date when the source hash it carries equals
    the tree's (not by file times: the .so travels to the GPU box prebuilt, and the hash is what proves it matches)."""
    lib = LIB.replace(".so", "_prof.so") if prof else LIB
    suffix = os.environ.get("LPCN_LIB_SUFFIX", "")          # experiments: LPCN_EXTRA_FLAGS=-D... LPCN_LIB_SUFFIX=x -> liblpcnet_hip_x.so, objects in build_x/
    if suffix:
        lib = LIB.replace(".so", "_" + suffix + ".so")
    pf = ["-DLPCN_ENABLE_PROF=1"] + (["-DLPCN_PROF_MASK=" + os.environ["LPCN_PROF_MASK"]] if "LPCN_PROF_MASK" in os.environ else []) if prof else []
    extra = os.environ.get("LPCN_EXTRA_FLAGS", "").split()
    h_src, h_dev = source_hashes(pf + extra)
    if not force and baked_hashes(lib) == (h_src, h_dev):
        return lib
    objdir = os.path.join(HERE, "build_" + suffix if suffix else ("build_prof" if prof else "build"))
    os.makedirs(objdir, exist_ok=True)
    objs = []

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    # engine.hip + the sample kernel's variants, one translation unit per streams-per-workgroup value: built in parallel
    jobs = [([HIPCC] + HIP_FLAGS + pf + extra + ["-c", os.path.join(CSRC, "engine.hip"), "-o", os.path.join(objdir, "engine.o")])]
    for sv in (1, 2, 4):
        jobs.append([HIPCC] + HIP_FLAGS + pf + extra + [f"-DLPCN_S={sv}", "-c", os.path.join(CSRC, "sample_variants.hip"), "-o", os.path.join(objdir, f"sample_s{sv}.o")])
    jobs.append([HIPCC] + HIP_FLAGS + pf + extra + ["-c", os.path.join(CSRC, "sample_x2.hip"), "-o", os.path.join(objdir, "sample_x2.o")])
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=5) as ex:
        list(ex.map(run, jobs))
    objs += [j[-1] for j in jobs]
    for c in ("api.c", "model_pack.c"):
        o = os.path.join(objdir, c[:-2] + ".o")
        # (the -D switches of an experiment reach the host C files too: lpcnet_engine.h's dealing defaults -- e.g. which waves of an int8 blob may
        # carry candidate heads -- are derived from the same macros the kernels are compiled with)
        run(["gcc"] + C_FLAGS + [x for x in extra if x.startswith("-D")] + [f'-DLPCN_SOURCE_HASH="{h_src}"', f'-DLPCN_DEVICE_SOURCE_HASH="{h_dev}"', "-c", os.path.join(CSRC, c), "-o", o])
        objs.append(o)
    run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs + ["-lpthread", "-lm"])
    return lib


def build_small_registry(verbose=False):
    """liblpcnet_hip_smallreg.so: the same objects with a model registry of 4 slots / 2 resident device sides (api.c rebuilt with
    -DLPCN_MAX_MODELS=4 -DLPCN_MAX_RESIDENT=2), so that tests reach the eviction paths that need > 256 distinct models in the product
    build (tests/test_gpu_wide.py; select with LPCNET_HIP_LIB).  Test artefact, git-ignored like every built library."""
    build(verbose=verbose)
    objdir0 = os.path.join(HERE, "build")
    needed = ("engine.o", "sample_s1.o", "sample_s2.o", "sample_s4.o", "sample_x2.o", "model_pack.o")
    if not all(os.path.exists(os.path.join(objdir0, f)) for f in needed):      # (a prebuilt library whose hash matches skips the compile: its objects may be absent -- ADVICE r5)
        build(force=True, verbose=verbose)
    lib = LIB.replace(".so", "_smallreg.so")
    h_src, h_dev = source_hashes()
    if baked_hashes(lib) == (h_src + "-smallreg", h_dev):
        return lib
    objdir = os.path.join(HERE, "build")
    o = os.path.join(objdir, "api_smallreg.o")
    cmds = [["gcc"] + C_FLAGS + ["-DLPCN_MAX_MODELS=4", "-DLPCN_MAX_RESIDENT=2", f'-DLPCN_SOURCE_HASH="{h_src}-smallreg"', f'-DLPCN_DEVICE_SOURCE_HASH="{h_dev}"',
                                 "-c", os.path.join(CSRC, "api.c"), "-o", o],
            [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + [os.path.join(objdir, f) for f in ("engine.o", "sample_s1.o", "sample_s2.o", "sample_s4.o", "sample_x2.o")]
            + [o, os.path.join(objdir, "model_pack.o"), "-lpthread", "-lm"]]
    for c in cmds:
        if verbose:
            print(" ".join(c), flush=True)
        subprocess.check_call(c)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, prof="--prof" in sys.argv))
