"""Build libsgpt_hip.so (gfx950) in-tree with hipcc.  No JIT, no torch extension machinery:
the C-ABI library has no torch types in it (include/sgpt_hip.h)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
# SGPT_EXPERIMENTS=1: the measurement scripts' build (libsgpt_hip_exp.so): environment A/B switches, the s_memtime stamps and
# the slower 32x32x16-MFMA re-tiling (gemm256w.hip) compiled in.  The product library has none of them.
EXPERIMENTS = os.environ.get("SGPT_EXPERIMENTS") == "1"
LIB = os.path.join(LIBDIR, "libsgpt_hip_exp.so" if EXPERIMENTS else "libsgpt_hip.so")
SOURCES = ["gemm.hip", "qgemm.hip", "gemm256q.hip", "attn.hip", "elementwise.hip", "topk.hip", "comm.hip", "api.hip"]
# the slower 32x32x16-MFMA re-tiling lives with the measurement scripts (scripts/micro/gemm256w.hip): experiment build only
EXTRA_SOURCES = [os.path.join(HERE, "..", "scripts", "micro", "gemm256w.hip")] if EXPERIMENTS else []
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wno-unused-result"] + (["-DSGPT_EXPERIMENTS"] if EXPERIMENTS else []) + os.environ.get("SGPT_EXTRA_FLAGS", "").split()
OBJ_SUFFIX = ".exp.o" if EXPERIMENTS else ".o"
# SGPT_LIB_TAG=name (with SGPT_EXTRA_FLAGS=-D...): a second build of the same ABI for same-box A/B runs (scripts/ab_libs.sh
# selects it through SGPT_HIP_LIB): libsgpt_hip_<name>.so, its own object files
if os.environ.get("SGPT_LIB_TAG"):
    LIB = os.path.join(LIBDIR, f"libsgpt_hip_{os.environ['SGPT_LIB_TAG']}.so")
    OBJ_SUFFIX = f".{os.environ['SGPT_LIB_TAG']}.o"


def _hipcc():
    rocm = os.environ.get("ROCM_PATH") or os.environ.get("ROCM_HOME")
    for c in (os.environ.get("HIPCC"), os.path.join(rocm, "bin", "hipcc") if rocm else None, "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def _rocm_libdir(hipcc):
    """Where librccl lives: $ROCM_PATH/lib, else next to the hipcc that compiles the sources (versioned prefixes such as
    /opt/rocm-7.2.0 work), else /opt/rocm/lib."""
    import shutil
    rocm = os.environ.get("ROCM_PATH") or os.environ.get("ROCM_HOME")
    cands = [os.path.join(rocm, "lib")] if rocm else []
    exe = hipcc if os.path.sep in hipcc else shutil.which(hipcc)
    if exe:
        cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(exe))), "lib"))
    cands.append("/opt/rocm/lib")
    for d in cands:
        if os.path.exists(os.path.join(d, "librccl.so")):
            return d
    return cands[-1]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    headers += [os.path.join(HERE, "..", "include", "sgpt_hip.h"), os.path.abspath(__file__)]
    objs, jobs = [], []
    for src in SOURCES + EXTRA_SOURCES:
        sp = src if os.path.isabs(src) or os.path.sep in src else os.path.join(CSRC, src)
        op = os.path.join(LIBDIR, os.path.basename(src).replace(".hip", OBJ_SUFFIX))
        objs.append(op)
        if force or _stale(op, [sp] + headers):
            jobs.append([hipcc] + FLAGS + ["-I", CSRC, "-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        # No -lrccl: comm.hip binds RCCL lazily (dlopen, preferring the copy already in the process -- torch ships its own
        # librccl.so.1 -- and checking its NCCL major version against the header it was compiled with).  A single-GPU user
        # needs no RCCL to build or load the library; rccl.h (types only) comes from the ROCm include directory.
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    try:
        build_host(force=force, verbose=verbose)
    except Exception as e:       # optional helper (the k = 1001 result dict in C): its failure must not fail the GPU library's build
        import warnings
        warnings.warn(f"host extension not built ({str(e).splitlines()[0]} ...): beir.assemble_results uses its Python form")
    return LIB


HOST_EXT = os.path.join(LIBDIR, "_sgpt_host.so")


def build_host(force=False, verbose=False):
    """The CPython helper of the host leg (csrc/host_assemble.c: the k = 1001 result dict in C), gcc against this
    interpreter's headers.  Optional: beir.assemble_results falls back to its Python form when the file is absent."""
    import shutil
    import sysconfig
    src = os.path.join(CSRC, "host_assemble.c")
    inc = sysconfig.get_paths()["include"]
    cc = os.environ.get("CC") or shutil.which("gcc") or shutil.which("cc")
    if not cc or not os.path.exists(os.path.join(inc, "Python.h")):
        return None
    if force or _stale(HOST_EXT, [src, os.path.abspath(__file__)]):
        cmd = [cc, "-O2", "-shared", "-fPIC", "-Wall", "-I", inc, src, "-o", HOST_EXT]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("host extension build failed:\n" + r.stdout + r.stderr)
    return HOST_EXT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
