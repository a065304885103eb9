// obj_loader.cpp — Wavefront OBJ ingest (C-ABI pt_load_obj).  Placeholder until the
// SURVEY.md §8(f) row 1 work lands: reports PT_ERR_UNSUPPORTED.
#include "../../include/pt_api.h"
void pt_set_error(const char* fmt, ...);
extern "C" int32_t pt_load_obj(const char* path, float scale, const float translate[3], PtPrimitive* prims, int32_t cap)
{
    (void)path; (void)scale; (void)translate; (void)prims; (void)cap;
    pt_set_error("pt_load_obj: not implemented yet");
    return PT_ERR_UNSUPPORTED;
}
