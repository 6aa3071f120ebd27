// zpq_vm.h -- ZPAQL interpreter shared by the kernels (HCOMP: contexts per coded byte) and by the
// host front end (PCOMP: the PostProcessor's PROG mode, decompressor.v:14-167).
//
// Reproduces ZPAQL.run / execute and the masked M/H accessors of the reference
// (zpaq/zpaql.v:167-211,215-954; oplen zpaq/types.v:51-64) including its quirks: operand
// fetch bounded by header.len (not hend), JT/JF/JMP offset ((N+128)&255)-127 applied after
// the operand fetch (one more than libzpaq, Q11), undefined opcodes end the run silently
// (Q12), division/modulo by zero are no-ops, shifts are masked to 5 bits, OUT only grows a
// host buffer: it is ignored on the device and appended to Vm::out on the host.  The reference has no step budget (a looping program
// hangs it); this one stops after ZPQ_VM_STEP_CAP steps and reports it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/zpaq_hip.h"

namespace zpqvm {

typedef int32_t i32;
typedef uint32_t u32;
typedef uint8_t u8;

struct Vm {
    u32 a, b, c, d;
    i32 f, pc;
    u8 *m; u32 mlen;
    u32 *h; u32 hlen;
    u32 *r;
    const u8 *hdr;
    i32 hdr_len, hbegin, hend;
    void *out;                   // host only: std::vector<uint8_t>* receiving OUT bytes (zpaql.v:151-159); null on the device
};

__host__ __device__ __forceinline__ u32 m_get(const Vm &z, u32 i) { return z.mlen ? z.m[i & (z.mlen - 1)] : 0u; }
__host__ __device__ __forceinline__ void m_set(Vm &z, u32 i, u32 v) { if (z.mlen) z.m[i & (z.mlen - 1)] = (u8)v; }
__host__ __device__ __forceinline__ u32 h_get(const Vm &z, u32 i) { return z.hlen ? z.h[i & (z.hlen - 1)] : 0u; }
__host__ __device__ __forceinline__ void h_set(Vm &z, u32 i, u32 v) { if (z.hlen) z.h[i & (z.hlen - 1)] = v; }

// operand columns 0..7 = A B C D *B *C *D N
__host__ __device__ inline u32 vm_src(const Vm &z, int s, u32 operand)
{
    switch (s) {
    case 0: return z.a;
    case 1: return z.b;
    case 2: return z.c;
    case 3: return z.d;
    case 4: return m_get(z, z.b);
    case 5: return m_get(z, z.c);
    case 6: return h_get(z, z.d);
    default: return operand;
    }
}
__host__ __device__ inline void vm_dst(Vm &z, int t, u32 v)
{
    switch (t) {
    case 0: z.a = v; break;
    case 1: z.b = v; break;
    case 2: z.c = v; break;
    case 3: z.d = v; break;
    case 4: m_set(z, z.b, v); break;
    case 5: m_set(z, z.c, v); break;
    default: h_set(z, z.d, v); break;
    }
}

// zpaql.v:167-175 + 215-954.  Returns false if the step cap was hit.
__host__ __device__ inline bool vm_run(Vm &z, u32 input)
{
    z.a = input;
    z.pc = z.hbegin;
    u32 steps = 0;
    while (z.pc < z.hend && z.pc >= z.hbegin) {
        u32 op = z.hdr[z.pc++];
        u32 operand = 0;
        const bool two = (op & 7) == 7 && op != 255;   // types.v:51-64
        if (two && z.pc < z.hdr_len) {
            operand = z.hdr[z.pc++];
        } else if (op == 255 && z.pc + 1 < z.hdr_len) {
            operand = z.hdr[z.pc] + z.hdr[z.pc + 1] * 256u;
            z.pc += 2;
        }
        const i32 rel = (i32)((operand + 128) & 255) - 127;  // jump quirk Q11
        bool go = true;
        if (op < 56) {
            const int t = op >> 3, k = op & 7;
            if (k == 0) {                  // X<>A (0 = NOP)
                if (t) { u32 tmp = vm_src(z, t, 0); vm_dst(z, t, z.a); z.a = tmp; }
            } else if (k == 1) vm_dst(z, t, vm_src(z, t, 0) + 1);
            else if (k == 2) vm_dst(z, t, vm_src(z, t, 0) - 1);
            else if (k == 3) vm_dst(z, t, ~vm_src(z, t, 0));
            else if (k == 4) vm_dst(z, t, 0);
            else if (k == 7) {
                if (t <= 3) vm_dst(z, t, z.r[operand & 255]);          // X=R N
                else if (t == 4) { if (z.f != 0) z.pc += rel; }        // JT
                else if (t == 5) { if (z.f == 0) z.pc += rel; }        // JF
                else z.r[operand & 255] = z.a;                         // R=A N
            } else go = false;             // 5,6,13,14,...: undefined, stops the run
        } else if (op < 64) {
            if (op == 56) go = false;                                             // HALT
            else if (op == 57) {                                                  // OUT (zpaql.v:151-159,382-384)
#if !defined(__HIP_DEVICE_COMPILE__)
                if (z.out) static_cast<std::vector<uint8_t> *>(z.out)->push_back((uint8_t)(z.a & 255u));
#endif
            }
            else if (op == 59) z.a = (z.a + m_get(z, z.b) + 512u) * 773u;         // HASH
            else if (op == 60) h_set(z, z.d, (h_get(z, z.d) + z.a + 512u) * 773u); // HASHD
            else if (op == 63) z.pc += rel;                                       // JMP
            else go = false;                                                      // 58,61,62
        } else if (op < 120) {
            vm_dst(z, (int)(op - 64) >> 3, vm_src(z, op & 7, operand));
        } else if (op < 128) {
            go = false;
        } else if (op < 216) {
            const u32 v = vm_src(z, op & 7, operand);
            switch ((op - 128) >> 3) {
            case 0: z.a += v; break;
            case 1: z.a -= v; break;
            case 2: z.a *= v; break;
            case 3: if (v) z.a /= v; break;
            case 4: if (v) z.a %= v; break;
            case 5: z.a &= v; break;
            case 6: z.a &= ~v; break;
            case 7: z.a |= v; break;
            case 8: z.a ^= v; break;
            case 9: z.a <<= (v & 31); break;
            default: z.a >>= (v & 31); break;
            }
        } else if (op < 240) {
            const u32 v = vm_src(z, op & 7, operand);
            const int g = (op - 216) >> 3;
            z.f = g == 0 ? (z.a == v) : (g == 1 ? (z.a < v) : (z.a > v));
        } else if (op == 255) {            // LJ
            if (z.pc < 2) go = false;
            else {
                z.pc = z.hbegin + (i32)z.hdr[z.pc - 2] + (i32)z.hdr[z.pc - 1] * 256;
                if (z.pc >= z.hend) go = false;
            }
        } else go = false;                 // 240..254
        if (!go) break;
        if (++steps >= ZPQ_VM_STEP_CAP) return false;
    }
    return true;
}

}  // namespace zpqvm
