// TEST INFRASTRUCTURE ONLY (oracle/): the XORWOW contract checked against rocRAND's OWN engine.
//
// SURVEY.md section 8c fixes the RNG contract of this repo as "XORWOW with rocRAND's seed scramble, subsequence 0, offset 0,
// uniform = 2.3283064e-10f + x * 2.3283064e-10f" (cuRAND's constants are not available offline; the reference seeds its subsequence
// from clock64() anyway).  The oracle (oracle/pt_oracle.cpp) and the kernels (csrc/pt_math.h: Rng) carry their own 20-line engine.
// This program runs rocRAND's real header-only engine (/opt/rocm/include/rocrand/rocrand_xorwow.h, rocrand_uniform.h: host + device
// code) on the HOST and prints its outputs, so that the restatement is pinned to the published implementation it claims to follow:
//   rocrand_ref SEED N   ->  N lines "raw uniform"   (rocrand_init(seed, 0, 0), then rocrand() / rocrand_uniform() on copies of the state)
// Built by `make -C oracle rocrand` with hipcc (no GPU is touched); tests/golden/ref_rocrand_xorwow.npz holds its outputs.
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_xorwow.h>
#include <rocrand/rocrand_uniform.h>
#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv)
{
    if (argc != 3) { fprintf(stderr, "usage: rocrand_ref SEED N\n"); return 1; }
    const unsigned long long seed = strtoull(argv[1], nullptr, 10);
    const int n = atoi(argv[2]);
    rocrand_state_xorwow st;
    rocrand_init(seed, /*subsequence*/ 0, /*offset*/ 0, &st);
    for (int i = 0; i < n; i++) {
        rocrand_state_xorwow c = st;                 // the uniform of the SAME draw, from a copy of the state
        const unsigned int r = rocrand(&st);
        const float u = rocrand_uniform(&c);
        unsigned int ub; __builtin_memcpy(&ub, &u, 4);
        printf("%u %u\n", r, ub);
    }
    return 0;
}
