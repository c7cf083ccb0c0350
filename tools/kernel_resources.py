#!/usr/bin/env python
"""Per-kernel register / scratch / LDS usage of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
    python tools/kernel_resources.py nemo_amd/csrc/gemm.hip [extra -I dirs...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(src, extra_inc=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-I",
           os.path.join(ROOT, "include"), "-I", os.path.dirname(os.path.abspath(src))]
    for i in extra_inc:
        cmd += ["-I", i]
    cmd += ["-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    res, cur = {}, None
    for line in out.splitlines():
        m = re.search(r"remark: [^ ]* *(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                      r"SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            m2 = re.search(r"Name: (\S+)", line)
            if m2:
                cur = res.setdefault(m2.group(1), {})
            continue
        if m.group(1) == "Function Name":
            cur = res.setdefault(m.group(2), {})
        elif cur is not None:
            cur[m.group(1).split(" [")[0]] = m.group(2)
    return res


if __name__ == "__main__":
    r = resources(sys.argv[1], sys.argv[2:])
    for k, v in r.items():
        name = k[:70]
        print(f"{name:70s} sgpr {v.get('TotalSGPRs','?'):>4} vgpr {v.get('VGPRs','?'):>4} agpr {v.get('AGPRs','?'):>4} scratch {v.get('ScratchSize','?'):>5} "
              f"sspill {v.get('SGPRs Spill','?'):>3} vspill {v.get('VGPRs Spill','?'):>3} occ {v.get('Occupancy','?')} lds {v.get('LDS Size','?')}")
