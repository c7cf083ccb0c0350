"""Static audit of the generated gfx950 code of every kernel (no GPU needed): instruction count, exec-mask regions
(`s_and_saveexec`: divergent control flow), quarter-rate integer multiplies (`v_mul_lo/hi_u32`), integer divisions
(`v_rcp_iflag_f32` sequences), fp divisions (`v_div_*`), VGPRs and scratch bytes.  `python tools/isa_audit.py [-D...] > file.md`"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nemo_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
extra = [a for a in sys.argv[1:] if a.startswith("-D")]

print("| file | kernel | instructions | exec-mask regions | v_mul_lo/hi_u32 | int div | fp div | VGPRs | scratch B |")
print("|---|---|---:|---:|---:|---:|---:|---:|---:|")
with tempfile.TemporaryDirectory() as tmp:
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") and f != "version.hip"):
        asm = os.path.join(tmp, src + ".s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-S", "--cuda-device-only",
                        "-I", os.path.join(ROOT, "include"), "-I", CSRC] + extra + [os.path.join(CSRC, src), "-o", asm],
                       check=True, stderr=subprocess.DEVNULL)
        txt = open(asm).read()
        scratch = dict(re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", txt))
        vgpr = dict(re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", txt))
        for k in re.split(r"\n(?=_Z[\w]+:\s)", txt):
            m = re.match(r"(_Z[\w]+):", k)
            if not m or m.group(1) not in vgpr:
                continue
            name = m.group(1)
            body = k.split(".Lfunc_end")[0]
            n = len([ln for ln in body.splitlines() if ln.startswith("\t") and not ln.strip().startswith((";", "."))])
            if "tap_reduce" in name and src != "convmod.hip":
                continue  # the shared helper is emitted into every translation unit
            print(f"| {src} | `{name[:64]}` | {n} | {body.count('s_and_saveexec')} | "
                  f"{body.count('v_mul_lo_u32') + body.count('v_mul_hi_u32')} | {body.count('v_rcp_iflag')} | "
                  f"{len(re.findall(r'v_div_fixup', body))} | {vgpr.get(name, '?')} | {scratch.get(name, '?')} |")
