#!/bin/bash
# tools/isa.sh <file.hip> [out.s] — device-only gfx950 assembly of one translation unit with the library's flags, plus a per-kernel
# summary (VGPRs, SGPRs, spills, LDS, code bytes) from the assembler's own metadata.  Runs without a GPU.
set -e
SRC=$1; OUT=${2:-/tmp/$(basename "$SRC" .hip).s}
DIR=$(cd "$(dirname "$0")/../tsar-mvs_amd/csrc" && pwd)
FLAGS=$(sed -n 's/^CXXFLAGS = //p' "$DIR/Makefile" | sed 's/\$(ARCH)/gfx950/')
/opt/rocm/bin/hipcc $FLAGS $EXTRA --cuda-device-only -S -o "$OUT" "$SRC"
python3 - "$OUT" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
# one metadata document at the end: .amdhsa_kernel blocks carry next_free_vgpr etc.; the yaml notes carry spill counts
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r"\." + k + r" (\S+)", body) or [None, "?"])[1]
    size = re.search(r"\.size\s+" + re.escape(name) + r", (\S+)", txt)
    print(f"{name[:150]}\n   vgpr {g('amdhsa_next_free_vgpr')} sgpr {g('amdhsa_next_free_sgpr')} accum_offset {g('amdhsa_accum_offset')} lds {g('amdhsa_group_segment_fixed_size')} scratch {g('amdhsa_private_segment_fixed_size')}")
for m in re.finditer(r"\.name:\s+(\S+).*?\.vgpr_spill_count:\s+(\d+)", txt, re.S):
    pass
PY
