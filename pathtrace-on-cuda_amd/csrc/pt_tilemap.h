// Which rank renders which 8x8 tile (host and device, include/pt_api.h: pt_render_tiles, pt_untile, pt_tile_map).
// Tiles are numbered row-major.  Every group of `world` consecutive tiles holds exactly one tile of every rank, and a rank's local tile
// `lt` lies in group `lt` — so a rank's tile buffer is the groups in order, whatever happens inside a group.  Inside group g the ranks are
// rotated by a hash of g (mode 2, default).  Mode 0 is the plain t % world of rounds 1-3, which for a frame whose tile row is a multiple
// of `world` (1920 / 8 = 240 tiles, world 8) gives each rank vertical stripes — the same columns in every row — and the ranks' times then
// differ by +-3 % with the content of their columns (r03_emulated_world.json: 0.392 ... 0.415 s); mode 1 rotates by the tile row of the
// group's first tile (diagonal stripes).  Pure scheduling: a pixel's samples do not depend on who renders them.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define PT_TM_HD __host__ __device__ inline
#else
#define PT_TM_HD inline
#endif

namespace ptd {

PT_TM_HD uint32_t tm_shift(uint32_t group, uint32_t world, uint32_t tiles_x, int mode)
{
    if (mode == 1) return (uint32_t)(((uint64_t)group * world / tiles_x) % world);
    if (mode == 2) { uint32_t h = group * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; return h % world; }
    return 0u;
}
// global tile of (rank, local tile); may be >= the number of tiles in the last, partial group
PT_TM_HD uint32_t tm_tile_of(uint32_t lt, uint32_t rank, uint32_t world, uint32_t tiles_x, int mode)
{
    const uint32_t sh = tm_shift(lt, world, tiles_x, mode);
    return lt * world + (rank + world - sh) % world;
}
PT_TM_HD void tm_owner(uint32_t tile, uint32_t world, uint32_t tiles_x, int mode, uint32_t& rank, uint32_t& lt)
{
    lt = tile / world;
    rank = (tile % world + tm_shift(lt, world, tiles_x, mode)) % world;
}

}  // namespace ptd
