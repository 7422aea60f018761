// match_device.h — the device half of cl_find_matches (match_kernels.hip), called from cl_match_api.cpp
#ifndef CL_MATCH_DEVICE_H
#define CL_MATCH_DEVICE_H

#include <stdint.h>

struct cl_context;

struct ClSuffixStats {
    uint32_t rounds = 0;      // prefix-doubling rounds (= rank levels kept for the LCP descent)
    float sort_ms = 0.f;      // suffix array: all rounds
    float lcp_ms = 0.f;       // LCP + inverse suffix array
};

// suffix array, LCP array (lcp[0] = 0, lcp[p] = LCP of the suffixes ranked p-1 and p) and inverse suffix array of
// text[0..n), whose last character must be the unique smallest one.  Host pointers in and out.
int cl_match_suffix_array(cl_context* ctx, const uint8_t* text, uint32_t n, uint32_t* sa, uint32_t* lcp, uint32_t* isa, ClSuffixStats* stats);

#endif
