"""Build libmvlt_hip.so (gfx950) in-tree with hipcc: one object per .hip file, compiled in parallel, then linked.

    python -m mvlt_amd.build [--force] [--verbose]
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libmvlt_hip.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]
# mlp.hip: the software-pipelined kernels interleave scalar-f32 activation code with MFMAs; hipcc's SLP vectoriser would pair it into
# v_pk_*_f32 again, and packed fp32 VALU does not overlap with MFMA on gfx950 (tools/probes/valu_rates.hip: 16 x {mfma + 4 v_pk_fma} runs
# at exactly the sum of the two, 16 x {mfma + 8 v_fma} hides half of the MFMA time)
EXTRA = {"mlp.hip": ["-fno-slp-vectorize"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources and the C ABI header: what a profile was collected FOR.  tools/roofline_traffic.py
    and tools/step_traffic.py stamp it into their outputs; bench.py drops a committed counter figure whose stamp is not the tree's."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc")))
    files.append(os.path.join(HERE, "..", "include", "mvlt_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(HERE, "..", "include", "mvlt_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force, verbose):
    s = os.path.join(CSRC, src)
    o = os.path.join(OBJ, src[:-4] + ".o")
    if not force and os.path.exists(o) and os.path.getmtime(o) >= max(os.path.getmtime(s), _deps_mtime()):
        return o
    cmd = ["hipcc", *FLAGS, *EXTRA.get(src, []), "-c", s, "-o", o]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return o


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, verbose), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = ["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or True))
