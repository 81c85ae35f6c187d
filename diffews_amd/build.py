"""Build libdiffews_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is a
plain C-ABI shared object (include/diffews_hip.h) loaded through ctypes."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdiffews_hip.so")
SOURCES = ["gemm.hip", "gemm_big.hip", "gemm8.hip", "conv_patch.hip", "conv_patch8.hip", "attention.hip", "vae_attention.hip", "attention_bwd.hip", "backward.hip", "norm.hip",
           "misc.hip", "preprocess.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result", "-Wno-unused-value"]


def _src_hash():
    """Content hash of every kernel source + the ABI header + the flags (mtimes do not survive the
    snapshot copy to the GPU box, contents do)."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, "..", "include", "diffews_hip.h")]
    for d in deps:
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read())
    return h.hexdigest()


def _stale():
    if not os.path.isfile(LIB) or not os.path.isfile(LIB + ".srchash"):
        return True
    with open(LIB + ".srchash") as f:
        return f.read().strip() != _src_hash()


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    if os.environ.get("DFW_NO_BUILD"):
        # e.g. under rocprofv3: the profiler's preloaded library has already initialised the GPU, and hipcc
        # children must not be started from such a process -- a stale library is an error there, not a rebuild
        raise RuntimeError(f"{LIB} is missing or older than its sources and DFW_NO_BUILD is set: "
                           "run `python -m diffews_amd.build` first")
    hipcc = os.environ.get("HIPCC", "hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    with open(LIB + ".srchash", "w") as f:
        f.write(_src_hash())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
