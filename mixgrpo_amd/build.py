"""Builds mixgrpo_amd/csrc/*.hip into libmixgrpo_hip.so for gfx950 with hipcc (in-tree, no torch extension).

`python -m mixgrpo_amd.build [--force]`.  hipcc cross-compiles without a GPU; the .so is git-ignored but
travels to the GPU box with the gpurun snapshot.

The in-tree library is ALWAYS built from the plain sources: no environment variable changes its flags.  Timing-only
diagnostic builds (`-DMGX_TIMING_ONLY_*`: wrong results by construction, csrc/common.h) are a separate artefact:
    python -m mixgrpo_amd.build --diagnostic scratch/libmixgrpo_diag.so -DMGX_TIMING_ONLY_NO_EPILOGUE
adds -DMGX_DIAGNOSTIC_BUILD (mgx_version() < 0, refused by mixgrpo_amd._lib.lib()), compiles into its own object
directory and may only be written under scratch/.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libmixgrpo_hip.so")
ARCH = "gfx950"

COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-Wno-unused-variable", "-Wno-unused-result", "-Wno-unused-value"]
# files whose results must be bit-identical to separately-rounded eager fp32 ops: no FMA contraction
# (norm.hip: the reference's RoPE / RMSNorm / LayerNorm are separately rounded eager fp32 ops too, and without the flag
# the two template instances of qk_norm_rope_fwd contract `y0*c0 - y1*s0` differently: the training forward and the
# recompute pass then disagree in isolated bf16 roundings)
PER_FILE = {"solver.hip": ["-ffp-contract=off"], "grpo.hip": ["-ffp-contract=off"], "norm.hip": ["-ffp-contract=off"]}


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0)")


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _generate():
    """Rewrite the generated instruction streams (csrc/gen/*.py -> csrc/*_body.inc); files are only touched on change."""
    sys.path.insert(0, os.path.join(CSRC, "gen"))
    try:
        import attn_bwd_dkv64
        import attn_bwd_dq64
        import attn_fwd64
        attn_fwd64.write()
        attn_bwd_dq64.write()
        attn_bwd_dkv64.write()
    finally:
        sys.path.pop(0)


def _digest(path, flags):
    h = hashlib.sha256()
    h.update(" ".join(flags).encode())
    for p in [path] + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".inc"))] + \
            [os.path.join(HERE, "..", "include", "mixgrpo_hip.h")]:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=True, diagnostic_out=None, diagnostic_flags=()):
    """Build the in-tree library (default), or -- with `diagnostic_out` -- a diagnostic one under scratch/."""
    obj_dir, lib_path, extra = OBJ, LIB, []
    if diagnostic_out is not None:
        lib_path = os.path.abspath(diagnostic_out)
        scratch = os.path.abspath(os.path.join(HERE, "..", "scratch"))
        if os.path.commonpath([lib_path, scratch]) != scratch:
            raise RuntimeError(f"diagnostic libraries may only be written under {scratch}")
        extra = ["-DMGX_DIAGNOSTIC_BUILD"] + list(diagnostic_flags)
        obj_dir = os.path.join(OBJ, "diag_" + hashlib.sha256(" ".join(extra).encode()).hexdigest()[:12])
    elif diagnostic_flags:
        raise RuntimeError("extra compile flags are only accepted for a diagnostic build (diagnostic_out=...)")
    os.makedirs(obj_dir, exist_ok=True)
    if diagnostic_out is None:
        # object directories of earlier diagnostic builds are dead weight in every gpurun snapshot (they travel with the tree)
        for name in os.listdir(OBJ):
            if name.startswith("diag_") and os.path.isdir(os.path.join(OBJ, name)):
                shutil.rmtree(os.path.join(OBJ, name), ignore_errors=True)
    _generate()
    hipcc = _hipcc()
    objs, jobs = [], []
    for src in _sources():
        flags = COMMON + extra + PER_FILE.get(src, [])
        path = os.path.join(CSRC, src)
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        stamp = obj + ".sha"
        dig = _digest(path, flags)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        jobs.append((src, [hipcc] + flags + ["-c", path, "-o", obj], stamp, dig))

    def run(job):
        src, cmd, stamp, dig = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-6000:]}")
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        with open(stamp, "w") as f:
            f.write(dig)
        return src

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for s in ex.map(run, jobs):
                if verbose:
                    print(f"[mixgrpo_amd.build] compiled {s}")
    if jobs or force or not os.path.exists(lib_path):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib_path] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[mixgrpo_amd.build] linked {lib_path}")
    return lib_path


if __name__ == "__main__":
    argv = sys.argv[1:]
    if "--diagnostic" in argv:
        i = argv.index("--diagnostic")
        build(force="--force" in argv, diagnostic_out=argv[i + 1],
              diagnostic_flags=[a for a in argv[i + 2:] if a.startswith("-D") or a.startswith("-f")])
    else:
        build(force="--force" in argv)
