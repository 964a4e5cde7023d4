#!/bin/bash
# ISA resource table of the shipped build: per kernel VGPRs / SGPRs / LDS / scratch (the kernel descriptors' numbers) and the frame of
# every device function that is a real call.   tools/isa_resources.sh > profiles/roundN_isa_resources.txt   (no GPU needed)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I $R/include --cuda-device-only -S $R/libmspack_amd/csrc/hip/shim.hip -o $T/shim.s 2>/dev/null
python3 - $T/shim.s $R <<'P'
import re, sys
s = open(sys.argv[1]).read()
print("# hipcc --offload-arch=gfx950 -O3 (ROCm 7.2), libmspack_amd/csrc/hip/shim.hip @ %s" % __import__("subprocess").run(["git", "-C", sys.argv[2], "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip())
print("%-26s %6s %6s %8s %9s %11s %11s" % ("kernel", "VGPRs", "SGPRs", "LDS B", "scratch B", "sgpr spills", "vgpr spills"))
for m in re.finditer(r"\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?\.sgpr_count:\s*(\d+).*?\.sgpr_spill_count:\s*(\d+).*?\.vgpr_count:\s*(\d+).*?\.vgpr_spill_count:\s*(\d+)", s, re.S):
    lds, name, scr, sg, sgs, vg, vgs = m.groups()
    nm = re.sub(r"^_Z\d+", "", name); nm = re.match(r"[a-z_0-9]+", nm).group(0)
    print("%-26s %6s %6s %8s %9s %11s %11s" % (nm, vg, sg, lds, scr, sgs, vgs))
print("\n# device functions that are real calls (a kernel's scratch = its own frame + the deepest chain of these):")
cur = None
for line in s.split("\n"):
    m = re.match(r"\s*\.type\s+(\S+),@function", line)
    if m: cur = m.group(1); code = vg = None; continue
    m = re.match(r"; codeLenInByte = (\d+)", line)
    if m: code = m.group(1)
    m = re.match(r"; NumVgprs: (\d+)", line)
    if m: vg = m.group(1)
    m = re.match(r"; ScratchSize: (\d+)", line)
    if m and cur and not re.match(r"_Z\d+mspack_", cur):
        mm = re.search(r"(lzx_pipe_[a-z_]+|qtm_update_model|lzx_copy_match_odd|zip_[a-z_]+|lzx_[a-z_]+)", cur)
        print("  %-28s code %6s B, %3s VGPRs, frame %4s B" % (mm.group(1) if mm else cur[:28], code, vg, m.group(1)))
        cur = None
P
rm -rf $T
