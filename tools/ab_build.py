"""Build library VARIANTS next to the default one, for same-box A/B runs on the GPU (box-to-box spread is +-1 ms per step,
larger than most single changes):

    python tools/ab_build.py base= pipe=-DMI355X_EXP_V4_PIPE incr=-DMI355X_EXP_EPI_INCR both="-DMI355X_EXP_V4_PIPE -DMI355X_EXP_EPI_INCR"
    python tools/ab_build.py base= noslp_attn=@attention:-fno-slp-vectorize      (a flag behind @<file stem>: applies to that translation unit only)

writes nemo_amd/lib_ab/libmi355x_asr_<name>.so (git-ignored, travels with the gpurun snapshot); on the GPU box
`tools/ab_run.sh` loops over them with MI355X_ASR_LIB=<variant> (parity subset, GEMM micro-benchmarks, bench.py)."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nemo_amd", "csrc")
OUT = os.path.join(ROOT, "nemo_amd", "lib_ab")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-I", os.path.join(ROOT, "include"),
        "-I", CSRC, "-Wno-unused-result"]


def build_variant(name, flags):
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        objs = []
        procs = []
        for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
            o = os.path.join(tmp, src[:-4] + ".o")
            objs.append(o)
            mine = [f.split(":", 1)[1] if f.startswith("@") else f for f in flags
                    if not f.startswith("@") or f[1:].split(":", 1)[0] == src[:-4]]
            procs.append((src, subprocess.Popen([HIPCC] + BASE + mine + ["-c", os.path.join(CSRC, src), "-o", o],
                                                stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        for src, pr in procs:
            out, _ = pr.communicate()
            if pr.returncode != 0:
                print(out)
                raise SystemExit(f"{name}: hipcc failed on {src}")
        lib = os.path.join(OUT, f"libmi355x_asr_{name}.so")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    print(f"[ab_build] {name}: {' '.join(flags) or '(default flags)'} -> {lib}")


if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    for arg in sys.argv[1:]:
        name, _, flags = arg.partition("=")
        build_variant(name, flags.split())
