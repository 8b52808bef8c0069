/* oracle/ocl_builtins.cl -- TEST INFRASTRUCTURE (like everything under oracle/): ROCm's OWN OpenCL builtin library behind
 * plain C names.
 *
 * What an OpenCL implementation's dot / cross / normalize / length and its sin / cos / cospi / sinpi round to is left to
 * the implementation (OpenCL 1.2 section 7.4).  The reference kernel (src/intersect_kernel.cl:65-127, 215-246, 284-304)
 * calls them, so "the reference on this GPU" is its source compiled against ROCm's library -- oracle/_ref/
 * ref_kernel_<tag>_strict.co, nothing standing in for anything.  This file lets a TEST-ONLY build of the HIP kernels
 * (-DOCRT_OCML_BUILTINS, opencl_raytracer_amd/lib_ocml) call the very same library functions: it is compiled by the same
 * clang, for the same target, with the strict build's floating-point options, to LLVM bitcode that already contains the
 * library's code (opencl.bc / ocml.bc are linked by the driver), and that bitcode is linked into the device side of
 * kernels.hip (-mlink-builtin-bitcode).  tests/test_ocml_pin.py then asks for ZERO differing float words between that
 * build and the strict code object: the HIP path's control flow and every formula of its own, checked against the
 * reference itself with no builder-written arithmetic in between.
 * Not part of the product: libocrt_hip.so is built without it and keeps the IEEE definitions of SURVEY.md 8a-0.3.
 *
 *   clang -x cl -cl-std=CL1.2 -Xclang -finclude-default-header -target amdgcn-amd-amdhsa -mcpu=gfx950 -O3 \
 *         -ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt -emit-llvm -c ocl_builtins.cl   (oracle/Makefile: ocl-builtins)
 */
float ocl_dot(float4 a, float4 b) { return dot(a, b); }
float4 ocl_cross(float4 a, float4 b) { return cross(a, b); }
float4 ocl_normalize(float4 v) { return normalize(v); }
float ocl_length(float4 v) { return length(v); }
float ocl_sin(float x) { return sin(x); }
float ocl_cos(float x) { return cos(x); }
float ocl_cospi(float x) { return cospi(x); }
float ocl_sinpi(float x) { return sinpi(x); }
float ocl_acos(float x) { return acos(x); }
