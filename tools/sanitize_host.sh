#!/bin/bash
# Host-only AddressSanitizer + UBSan run of the CPU oracle and of the library's host-side code
# (BVH build, sky cooking, generators, scene-file loader incl. its TOML / OBJ / PNG parsers on mutated inputs).  GPU sanitizers are not available on this pool.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); D=${PT_SAN_DIR:-/tmp/asan}; mkdir -p $D
make -s -C $R/oracle tables.o && make -s -C $R/pathtracer_amd/csrc tables.o
F="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared -pthread"
g++ $F $R/oracle/pt_oracle.cpp $R/oracle/tables.o -o $D/libptoracle.so
cat > $D/stub.cpp <<'EOS'
// stand-ins for the two error helpers that live in core.hip (GPU side, not part of this build)
#include <string>
struct pt_ctx;
static std::string g_err;
int ptc_fail(pt_ctx*, int code, const std::string& msg) { g_err = msg; return code; }
extern "C" const char* pt_last_error(const pt_ctx*) { return g_err.c_str(); }
EOS
g++ $F $R/pathtracer_amd/csrc/host_bvh.cpp $R/pathtracer_amd/csrc/host_sky.cpp $R/pathtracer_amd/csrc/host_gen.cpp $R/pathtracer_amd/csrc/host_scene_io.cpp $D/stub.cpp $R/pathtracer_amd/csrc/tables.o -o $D/libpthost.so
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so) $LD_PRELOAD" ASAN_OPTIONS=detect_leaks=0 PT_SAN_DIR=$D python3 $R/tools/sanitize_host_run.py
# ThreadSanitizer over the threaded BVH builder (subtrees build in parallel)
cat > $D/tsan_drv.cpp <<'EOS'
#include <cstdio>
#include <vector>
#include "ptcore.h"
int main() {
    for (uint32_t n : {1u, 2u, 1000u, 200000u, 1000000u}) {
        std::vector<pt_triangle> t(n);
        const float lo[3] = {-1, -1, -4}, hi[3] = {1, 1, -2};
        if (pt_gen_soup(0x50554F53ull, n, lo, hi, 0.05f, 0.0f, 8, t.data())) return 1;
        std::vector<pt_bvh_node> nodes(2 * (size_t)n - 1);
        uint32_t cnt = 0;
        if (pt_bvh_build(t.data(), n, nodes.data(), &cnt) || cnt != 2 * n - 1) return 2;
    }
    std::puts("thread sanitizer run clean");
    return 0;
}
EOS
g++ -O1 -g -std=c++17 -ffp-contract=off -fsanitize=thread -pthread -I$R/include $D/tsan_drv.cpp $R/pathtracer_amd/csrc/host_bvh.cpp $R/pathtracer_amd/csrc/host_gen.cpp -o $D/tsan_drv
$D/tsan_drv
