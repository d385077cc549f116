/* vgen_hip_hooks.h — test-only entry points of tests/native/libvgen_hip_hooks.so: the product's sources compiled a second
 * time with -DVGEN_TEST_HOOKS.  Nothing here exists in the shipped vgen_amd/libvgen_hip.so
 * (tests/test_abi_and_sharding.py asserts the symbol's absence); include/vgen_hip.h does not declare it.
 *
 * Also only in that build: the environment variable VGEN_DEBUG_GTAB_FAIL=<bits> — wide generator tables of <bits> bits or
 * more behave as if their allocation had failed (the context must step down, runtime.cpp: ensure_gtab). */
#ifndef VGEN_HIP_HOOKS_H
#define VGEN_HIP_HOOKS_H
#include "../../include/vgen_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Makes the context's dispatches fail with VGEN_E_HIP after `after_dispatches` more of them have been accepted (as a device
 * that drops off the bus would); UINT64_MAX disarms.  The reference has no fault injection; the multi-device
 * failure semantics of vgen_scan_multi are tested through this. */
int vgen_debug_fail_after(vgen_ctx *ctx, uint64_t after_dispatches);
#ifdef __cplusplus
}
#endif
#endif
