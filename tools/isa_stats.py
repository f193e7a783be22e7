"""Static per-kernel numbers from the gfx950 assembly of one .hip file: instructions, VGPRs, SGPRs, LDS, scratch, spills.  What a change
does to a kernel's hot path can be read here before a GPU is at hand.   python tools/isa_stats.py mf_odometry.hip [kernel-substring]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "maskfusion_amd", "csrc")
sys.path.insert(0, ROOT)
from maskfusion_amd.build import FILE_FLAGS  # noqa: E402


def stats(src):
    asm = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *FILE_FLAGS.get(src, []), "-S", "--cuda-device-only", "-o", "-",
                          os.path.join(CSRC, src)], capture_output=True, text=True, check=True).stdout
    out = {}
    for m in re.finditer(r"^(_Z\w+|k_\w+):[^\n]*\n(.*?)^\s*s_endpgm", asm, re.S | re.M):
        body = m.group(2)
        n = sum(1 for l in body.split("\n") if re.match(r"^\s+(v_|s_|ds_|buffer_|global_|flat_|scratch_)", l))
        out[m.group(1)] = dict(instructions=n, calls=len(re.findall(r"s_swappc_b64", body)))
    for m in re.finditer(r"- \.agpr_count:.*?\.name:\s+(\S+)(.*?)\.wavefront_size", asm, re.S):
        pass
    for blk in re.split(r"\n  - ", asm[asm.find("amdhsa.kernels"):]):
        nm = re.search(r"\.name:\s+(\S+)", blk)
        if not nm or nm.group(1) not in out:
            continue
        for key in ("vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count"):
            v = re.search(r"\." + key + r":\s+(\d+)", blk)
            out[nm.group(1)][key] = int(v.group(1)) if v else None
    return out


if __name__ == "__main__":
    src = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    for k, v in stats(src).items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        if sub in name:
            print(f"{name[:64]:64s} instr {v['instructions']:5d} calls {v['calls']} vgpr {v.get('vgpr_count')} sgpr {v.get('sgpr_count')} "
                  f"lds {v.get('group_segment_fixed_size')} scratch {v.get('private_segment_fixed_size')} spills {v.get('vgpr_spill_count')}")
