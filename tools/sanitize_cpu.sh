#!/bin/bash
# tools/sanitize_cpu.sh — the CPU side under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on the
# pool; SURVEY section 5 "race detection / sanitizers": the CPU restatement and the host code are what can be checked this way).
#   1. the oracle (oracle/*.c) rebuilt with -fsanitize=address,undefined, driven by its CPU tests (known answers, the reference-compiled
#      SLIC fixtures, the independent float64 restatement, the golden vectors);
#   2. the C++ host tool (host/tsar_gipuma.cpp + tsar_io.h + tsar_jpeg.h: PGM / PPM / PNG / JPEG / .dmb / cam / pair parsers, the
#      resume logic) rebuilt the same way, driven by the CPU half of tests/test_io_cli.py and tests/test_jpeg_decode.py.
# Both builds go to a scratch directory and replace the real artefacts only for the duration of the run.  Exit status 0 = clean.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
S=$(mktemp -d)
trap 'cp "$S/oracle.bak" "$ROOT/oracle/libtsar_oracle.so"; cp "$S/cli.bak" "$ROOT/tsar-mvs_amd/tsar_gipuma"; touch "$ROOT/oracle/libtsar_oracle.so"; rm -rf "$S"' EXIT
cp "$ROOT/oracle/libtsar_oracle.so" "$S/oracle.bak"; cp "$ROOT/tsar-mvs_amd/tsar_gipuma" "$S/cli.bak"
(cd "$ROOT/oracle" && gcc -O1 -g -ffp-contract=off -mfma -mavx2 -fopenmp -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
     -o "$ROOT/oracle/libtsar_oracle.so" tsar_oracle.c tsar_oracle_fusion.c tsar_oracle_slic.c tsar_oracle_texture.c -lm)
touch "$ROOT/oracle/libtsar_oracle.so"
cd "$ROOT"
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 OMP_NUM_THREADS=4 \
    python -m pytest tests/test_oracle_known_answers.py tests/test_slic_reference_golden.py tests/test_oracle_independent_float64.py tests/test_oracle_independent_sweep.py tests/test_golden.py -x -q -m "not gpu"
cp "$S/oracle.bak" "$ROOT/oracle/libtsar_oracle.so"; touch "$ROOT/oracle/libtsar_oracle.so"
(cd "$ROOT/tsar-mvs_amd" && g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -pthread -o tsar_gipuma host/tsar_gipuma.cpp -L. -ltsar_hip -lz -Wl,-rpath,"$ROOT/tsar-mvs_amd")
ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_io_cli.py tests/test_jpeg_decode.py -q -m "not gpu"
#   3. the JPEG reader (host/tsar_jpeg.h) on damaged files: tools/jpeg_fuzz.cpp, 3000 mutations each of a baseline 4:2:0, a
#      progressive 4:2:2, a 4:4:4 file with restart markers and optimised tables, and a gray file (written here with Pillow)
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -o "$S/jpeg_fuzz" "$ROOT/tools/jpeg_fuzz.cpp"
python - "$S" <<'PY'
import sys
import numpy as np
from PIL import Image
a = np.random.default_rng(0).integers(0, 256, (67, 91, 3)).astype(np.uint8)
Image.fromarray(a).save(sys.argv[1] + "/f1.jpg", quality=80, subsampling=2)
Image.fromarray(a).save(sys.argv[1] + "/f2.jpg", quality=80, subsampling=1, progressive=True)
Image.fromarray(a).save(sys.argv[1] + "/f3.jpg", quality=95, subsampling=0, restart_marker_blocks=3, optimize=True)
Image.fromarray(a[..., 0]).save(sys.argv[1] + "/f4.jpg", quality=60)
PY
TMPDIR="$S" "$S/jpeg_fuzz" 3000 "$S/f1.jpg" "$S/f2.jpg" "$S/f3.jpg" "$S/f4.jpg"
echo "sanitize_cpu: clean"
