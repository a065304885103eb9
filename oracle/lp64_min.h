// TEST INFRASTRUCTURE ONLY (oracle/).  Force-included when compiling the reference's
// srcs/bvh.cpp for oracle/_ref.
//
// srcs/bvh.cpp:204 (BVH::Cluster_Select_K, the deprecated K-means path that nothing on
// the render path calls) evaluates `min((unsigned long long)(2*K), vector::size() - ...)`.
// On MSVC x64 both arguments are `unsigned long long`; on LP64 Linux size_t is
// `unsigned long`, so no overload matches and the translation unit does not compile.
// This single overload settles that integer-width difference.  It is not a stand-in for
// any header, library or tool, and the function that uses it is never executed here.
#pragma once
#ifdef __cplusplus
static inline unsigned long long min(unsigned long long a, unsigned long b)
{
    return a < (unsigned long long)b ? a : (unsigned long long)b;
}
#endif
