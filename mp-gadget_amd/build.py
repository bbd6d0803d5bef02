"""Build the gfx950 engine (hipcc cross-compiles without a GPU): mp-gadget_amd/libmpgadget_hip.so.

What is rebuilt is decided by CONTENT, not by file times: every object carries the SHA-256 of its source, of all headers and of the
compiler flags (csrc/_obj/<name>.o.hash), and the library exports the hash over all of them (mpg_build_stamp()).  engine.py compares that
stamp with the sources it finds next to the library and refuses a stale one, so a library that does not match the tree cannot be
loaded silently - here, on the GPU box (the built .so travels with the snapshot) or on the driver's box."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libmpgadget_hip.so")
SOURCES = ["tree_build.hip", "grav_walk.hip", "grav_walk_coop.hip", "grav_walk_split.hip", "grav_pair_walk.hip", "pm.hip", "sph.hip", "timestep.hip", "peano.hip", "domain.hip", "fof.hip", "snapshot_io.hip", "engine.hip", "dist.hip", "rccl_comm.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
# experiments (-DMPG_EXP_...): MPG_EXTRA_FLAGS="file.hip:-Dx -Dy" applies to one source, MPG_EXTRA_FLAGS="-Dx" to all; part of the
# build stamp like every flag
_EXTRA = os.environ.get("MPG_EXTRA_FLAGS", "")
_EXTRA_FILE, _EXTRA = (_EXTRA.split(":", 1) if ".hip:" in _EXTRA else ("", _EXTRA))


def _headers():
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    hs.append(os.path.join(HERE, "..", "include", "mpgadget_hip.h"))
    return hs


def _sha(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def source_stamp():
    """the stamp a library built from the sources of this tree carries"""
    return _sha([os.path.join(CSRC, s) for s in SOURCES] + _headers(), " ".join(FLAGS) + os.environ.get("MPG_EXTRA_FLAGS", ""))[:32]


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return ""


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = _headers()

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        flags = FLAGS + (_EXTRA.split() if _EXTRA_FILE in ("", src) else [])
        want = _sha([s] + headers, " ".join(flags))
        if force or not os.path.exists(o) or _read(o + ".hash") != want:
            cmd = [HIPCC] + flags + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            with open(o + ".hash", "w") as f:
                f.write(want)
            return o, True
        return o, False

    with ThreadPoolExecutor(max_workers=4) as ex:
        res = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in res]
    stamp = source_stamp()
    if force or any(ch for _, ch in res) or not os.path.exists(LIB) or _read(LIB + ".stamp") != stamp:
        sc = os.path.join(OBJ, "build_stamp.c")
        with open(sc, "w") as f:
            f.write('const char *mpg_build_stamp(void) { return "%s"; }\n' % stamp)
        so = os.path.join(OBJ, "build_stamp.o")
        subprocess.check_call(["gcc", "-O1", "-fPIC", "-c", sc, "-o", so])
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + [so] + \
              ["-L/opt/rocm/lib", "-lrocfft", "-ldl", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        with open(LIB + ".stamp", "w") as f:
            f.write(stamp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
