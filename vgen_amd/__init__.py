"""vgen_amd — MI355X-native scan engine for the vanity-address hot path of oritwoen/vgen.

The product is vgen_amd/libvgen_hip.so (HIP kernels for gfx950 + host runtime behind the C ABI of
include/vgen_hip.h).  This package is the thin Python view of that ABI used by the tests and by
bench.py; names mirror the reference (src/address.rs, src/scanner.rs, src/pattern.rs, src/gpu.rs).
There is no CPU fallback: importing works anywhere, creating a GpuRunner needs an MI355X.
"""
from .api import (AddressFormat, GeneratedAddress, GpuRunner, Pattern, ScanConfig, ScanResult, VgenError,
                  abi_version, address_from_payload, derive, device_count, device_name, key_add, key_variant, key_to_wif, random_key,
                  library_path, scan_gpu_with_runner, ProviderResult, provider_resolve, build_pattern, build_exact_pattern)

__all__ = ["AddressFormat", "GeneratedAddress", "GpuRunner", "Pattern", "ScanConfig", "ScanResult", "VgenError",
           "abi_version", "address_from_payload", "derive", "device_count", "device_name", "key_add", "key_variant",
           "key_to_wif", "random_key", "library_path", "scan_gpu_with_runner", "ProviderResult", "provider_resolve", "build_pattern",
           "build_exact_pattern"]
