// Host stand-in for <hip/hip_runtime.h>, only for tests/box_logic.cpp: lets raytracer_amd/csrc/rt_device.h and
// rt_box.h (plain arithmetic, no wave-level code) compile with clang++ for the CPU.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline long long __double_as_longlong(double d) { long long u; std::memcpy(&u, &d, 8); return u; }
static inline double __longlong_as_double(long long u) { double d; std::memcpy(&d, &u, 8); return d; }
