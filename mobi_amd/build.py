"""Build the gfx950 engine library (libmobi_hip.so) in-tree with hipcc.

    python -m mobi_amd.build            # build if sources are newer than the library
    python -m mobi_amd.build --force

hipcc cross-compiles for gfx950 without a GPU present.  The library is kept in
`mobi_amd/` (git-ignored, but it travels with the source snapshot to the GPU box).
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmobi_hip.so")
OBJ = os.path.join(HERE, "csrc", "_obj")
ARCH = "gfx950"

# per-file extra flags; sampler arithmetic is kept un-contracted (no FMA fusion) so the
# fp32 latent update is bit-identical to the reference's separate mul / add ops
EXTRA = {"sampler_ops.hip": ["-ffp-contract=off"], "postprocess.hip": ["-ffp-contract=off"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "mobi_engine.h"))
    return max(os.path.getmtime(d) for d in deps)


STAMP = LIB + ".flags"


def sources_sha16():
    """sha256 over what the library is built from (every .hip / .h of csrc/, include/mobi_engine.h, the flags): the identity of a
    build that survives a rebuild in another directory (the binary's own hash need not)."""
    import hashlib
    h = hashlib.sha256()
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    deps.append(os.path.join(os.path.dirname(HERE), "include", "mobi_engine.h"))
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(_flags_id().encode())
    return h.hexdigest()[:16]


def _flags_id():
    """What the library was compiled with besides the sources: a library built with A/B or debug flags
    (`MOBI_HIPCC_FLAGS=-DMOBI_DBG_...`, timing-only variants with WRONG results) must never pass as current
    for a plain build, whatever its mtime says."""
    return repr((ARCH, os.environ.get("MOBI_HIPCC_FLAGS", "").split(), sorted(EXTRA.items())))


def up_to_date():
    if not (os.path.exists(LIB) and os.path.exists(STAMP) and os.path.getmtime(LIB) >= _deps_mtime()):
        return False
    with open(STAMP) as f:
        return f.read() == _flags_id()


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MObI engine needs the ROCm toolchain to build")
    return exe


def build(force=False, verbose=True):
    if not force and up_to_date():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    common = [cc, f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"]
    common += os.environ.get("MOBI_HIPCC_FLAGS", "").split()

    def compile_one(src):
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        cmd = common + EXTRA.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if os.path.exists(obj):
            os.remove(obj)                      # a failed compile must never leave yesterday's object to be linked
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(obj) or "error:" in r.stderr:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, sources()))
    tmp = LIB + ".tmp"
    r = subprocess.run([cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    os.replace(tmp, LIB)
    with open(STAMP, "w") as f:
        f.write(_flags_id())
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB) from {len(objs)} sources")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
