// filter.h — vgen_filter: a compiled pattern = exact DFA (host confirmation) + device prefilter.
//
// The reference evaluates Pattern::matches on the host for every key of every batch
// (src/gpu.rs:1030-1093).  Here the pattern is analysed once: from its DFA we derive a NECESSARY
// condition on the 20-byte payload that the kernel can test in a few instructions —
//   Base58 formats : the accepted address prefixes become hash160 ranges (big-integer bounds of
//                    "prefix * 58^k" for every feasible address length),
//   Bech32 / hex   : accepted leading symbols and required trailing symbols become bit masks over
//                    the payload (and over the Bech32 checksum, which the kernel recomputes),
// and every candidate the device reports is confirmed on the host with the exact DFA over the
// encoded address string, so results equal the reference's host-side filter bit for bit.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../device/device_types.h"
#include "regex_dfa.h"

struct vgen_filter {
    std::string pattern;
    bool case_insensitive = false;
    uint32_t format = 0;
    vg::Dfa dfa;            // decides Pattern::matches exactly
    vg::DevFilter dev{};    // device prefilter (superset)
    double selectivity = 1.0;   // estimated fraction of keys the device reports
    std::vector<uint32_t> chk_lut;   // Bech32 checksum tables (20 x 256) when the prefilter tests the checksum
    std::vector<uint32_t> dfa_blob;  // DEVF_DFA: the DFA in device layout (core/dfa_eval.h)
};

namespace vg {

// Compiles pattern + derives the device prefilter for `format`. false + err on invalid patterns.
bool filter_compile(const std::string &pattern, bool case_insensitive, uint32_t format, vgen_filter &out,
                    std::string &err);

}  // namespace vg
