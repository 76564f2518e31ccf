#!/bin/bash
# AddressSanitizer + UBSan and ThreadSanitizer runs of the HOST part of the library (symbolic analysis on 1 / 3 / 8
# threads, lazy CSR pattern, mesh refinement, eigensolver) on the C1 meshes.  CPU only; GPU sanitizers are not
# available on the pool.  usage: scripts/sanitize_host.sh   (from the repo root)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/plfem_sanitize && mkdir -p $OUT
cd $ROOT
python3 - <<PY
import numpy as np
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh
g = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0)
for lv in (0, 1):
    m = generate_mesh(g, 1.0, lv)
    p = np.ascontiguousarray(m.p, dtype=np.float64); t = np.ascontiguousarray(m.t, dtype=np.int32)
    with open('$OUT/mesh%d.bin' % lv, 'wb') as f:
        np.array([p.shape[1], t.shape[1]], dtype=np.int32).tofile(f); p.tofile(f); t.tofile(f)
PY
SRC="pl_fem_vectoriel_amd/csrc/symbolic.cpp pl_fem_vectoriel_amd/csrc/plan.cpp pl_fem_vectoriel_amd/csrc/host_eig.cpp pl_fem_vectoriel_amd/csrc/api_host.cpp pl_fem_vectoriel_amd/csrc/api_debug_host.cpp scripts/micro/sanitize_host.cpp"
for san in address,undefined thread; do
  g++ -std=c++17 -O1 -g -fsanitize=$san -fno-omit-frame-pointer -pthread -DPLFEM_TEST_HOOKS=1 -Iinclude -Ipl_fem_vectoriel_amd/csrc $SRC -o $OUT/h_${san%%,*}
done
ASAN_OPTIONS=detect_leaks=1 $OUT/h_address $OUT/mesh0.bin $OUT/mesh1.bin
TSAN_OPTIONS=halt_on_error=1 $OUT/h_thread $OUT/mesh0.bin $OUT/mesh1.bin
echo "sanitizers clean"
