"""Build libkrisp_hip.so (hand-written HIP for gfx950) in-tree with hipcc."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "krisp_hip.hip")
INC = os.path.join(ROOT, "include")
LIB = os.path.join(HERE, "libkrisp_hip.so")


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def source_sha16():
    """names the build: sha256 over the HIP sources and the header (what profiles/traffic.json says it was measured with)"""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.dirname(SRC)
    for f in sorted(os.listdir(csrc)) + [os.path.join(INC, "krisp_hip.h")]:
        path = f if os.path.isabs(f) else os.path.join(csrc, f)
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def build(force=False, verbose=False):
    csrc = os.path.dirname(SRC)          # krisp_hip.hip #includes its parts (k_*.inc kernels, h_*.inc host)
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(INC, "krisp_hip.h")]
    if (not force and os.path.exists(LIB)
            and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps)):
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           f"-I{INC}", "-o", LIB, SRC, "-lrccl", "-lz", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
