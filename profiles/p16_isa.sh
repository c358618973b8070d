#!/bin/bash
# usage: bash profiles/p16_isa.sh "<demangled kernel substring>" [extra -D flags]  -> /tmp/p16_isa.s (instructions only) of that kernel (fast build)
set -e
cd "$(dirname "$0")/../attention-gan_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DAGAN_P16_FAST_BUILD $2 -c conv_p16.hip -o /tmp/p16_fast.o
cd /tmp && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading p16_fast.o > /dev/null 2>&1
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn p16_fast.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 2>/dev/null | sed 's/ *\/\/.*//' > /tmp/p16_all.s
rm -f p16_fast.o.0.*
python3 - "$1" <<'P'
import re, subprocess, sys
lines = open("/tmp/p16_all.s").read().split("\n")
starts = [(i, l) for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <_ZN", l)]
for k, (i, l) in enumerate(starts):
    dem = subprocess.run(["c++filt", l.split("<")[1].rstrip(">:")], capture_output=True, text=True).stdout
    if sys.argv[1] in dem:
        end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        open("/tmp/p16_isa.s", "w").write("\n".join(x.strip() for x in lines[i:end]))
        print(dem.strip()[:120], end - i, "lines")
        break
P
