"""Builds libdfine_hip.so (every csrc/*.hip, gfx950 only) in-tree with hipcc.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot; hipcc
cross-compiles without a GPU, so `build()` is also the CPU-side "does it build" check.
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
INCLUDE_DIR = os.path.join(os.path.dirname(PKG_DIR), "include")
LIB_PATH = os.path.join(CSRC_DIR, "libdfine_hip.so")

HIPCC_FLAGS = [
    "-O3",
    "-std=c++17",
    "--offload-arch=gfx950",
    "-munsafe-fp-atomics",  # hardware global_atomic_add_f32, never a CAS loop
    "-fPIC",
    "-shared",
]


def sources():
    return sorted(
        os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith(".hip")
    )


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = sources() + [
        os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith(".h")
    ]
    deps += [os.path.join(INCLUDE_DIR, f) for f in os.listdir(INCLUDE_DIR)]
    return any(os.path.getmtime(d) > built for d in deps)


def find_hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libdfine_hip.so cannot be built")


def _object_path(src):
    return os.path.join(CSRC_DIR, "build", os.path.basename(src)[:-4] + ".o")


def _compile_one(args):
    src, obj, verbose = args
    cmd = [find_hipcc(), *[f for f in HIPCC_FLAGS if f != "-shared"], "-c", f"-I{INCLUDE_DIR}", f"-I{CSRC_DIR}", "-o", obj, src]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    return src, proc.returncode, proc.stdout + proc.stderr


def build(force=False, verbose=False):
    """Compile csrc/*.hip into libdfine_hip.so if it is missing or older than its sources: one object per
    source (only the stale ones, in parallel), then one link."""
    if not force and not _stale():
        return LIB_PATH
    import fcntl

    os.makedirs(os.path.join(CSRC_DIR, "build"), exist_ok=True)
    # one builder at a time: the N ranks of `bench.py --gpus N` / torchrun all come through here at once, and two hipcc runs
    # writing the same objects / the same .so would hand a third rank a half-written library. The others wait, then find it fresh
    lock = open(os.path.join(CSRC_DIR, "build", ".lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and not _stale():
            return LIB_PATH
        return _build_locked(force, verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(force, verbose):
    from concurrent.futures import ThreadPoolExecutor

    headers = [os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith(".h")]
    headers += [os.path.join(INCLUDE_DIR, f) for f in os.listdir(INCLUDE_DIR)]
    newest_header = max(os.path.getmtime(h) for h in headers)
    jobs = []
    for src in sources():
        obj = _object_path(src)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
            jobs.append((src, obj, verbose))
    with ThreadPoolExecutor(max_workers=min(8, max(len(jobs), 1))) as pool:
        for src, rc, log in pool.map(_compile_one, jobs):
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {src} ({rc}):\n{log}")
    tmp = LIB_PATH + f".tmp{os.getpid()}"
    cmd = [find_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *[_object_path(s) for s in sources()]]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc link failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    os.replace(tmp, LIB_PATH)  # atomic: a process loading the library never sees a half-written file
    return LIB_PATH
