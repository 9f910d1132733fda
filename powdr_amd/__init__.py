"""powdr_amd — MI355X-native backend for powdr's OpenVM autoprecompile proving path.

The product is `libpowdr_gpu.so` (HIP kernels + C ABI, see include/*.h); this
package only loads it (ctypes) and mirrors the reference's host-side interface
for tests and benchmarks. There is no CPU fallback: importing `powdr_amd.abi`
raises if the HIP library is missing.
"""
__all__ = ["abi", "synth", "build"]
