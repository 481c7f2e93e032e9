#!/usr/bin/env python3
"""Generates multihop_dense_retrieval_amd/csrc/mdr_encoder_gemm_quad_loop.inc: the hand-scheduled K-loop of gemm_quad_kernel
(256x256x64 f16 tile, four waves of 128x128, one wave per SIMD) as inline-asm text with fixed registers.

    python scripts/gen_gemm_quad_asm.py [--sched 3,3,3,3,2,2,0,0] > multihop_dense_retrieval_amd/csrc/mdr_encoder_gemm_quad_loop.inc

Why generated asm: hipcc cannot allocate this loop (256 accumulator + 128 fragment registers of the wave's 512; it answers with 25-500 spilled
registers whatever the source looks like -- DESIGN.md section 4), and a single wave per SIMD needs every non-MFMA instruction placed in the
shadow of an MFMA by hand. The skeleton was first measured alone (scripts/ubench/quad_loop.hip, profiles/r03_quad_loop_ubench.txt).

Register map (per wave; the clobber list of the asm statements):
  a[0:255]   accumulators, tile (m, n) at a[4 (8 m + n)]  -- live ACROSS the asm statements: the epilogue reads them with v_accvgpr_read
  v[0:31]    A fragments of k-half 0, v[32:63] W fragments of k-half 0, v[64:95] / v[96:127] the same for k-half 1
  v128/v129  LDS read address of A / W, k-half 0 (slot included); v130/v131 k-half 1
  v[132:139] buffer offsets of the wave's 8 A pieces, v[140:147] of its 8 W pieces
  s20 / s21  scalar buffer offset of the loader's K-tile in A (K position only: the rows are in v[132:139]) / W; s24 loop counter; s25 LDS destination of piece 0 (wave and slot included)
One step = one K-tile: phase 0 (64 MFMAs on k-half 0; reads k-half 1 of the same slot) | s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier (slot free,
next K-tile landed) | phase 1 (64 MFMAs on k-half 1; reads k-half 0 of the next K-tile from the other slot). The DMA pieces of K-tile T+2 are
issued right behind the barrier of step T (`sched`: pieces per 16-MFMA sub-phase, 4 of phase 1 then 4 of the next phase 0).
A statement covers ONE output tile of KT K-tiles (KT even, >= 4): step 0 writes the accumulators with C = 0; steps 1..KT-3 loop; step KT-2 moves
the loader to the NEXT tile (whose K-tile 0 and the first pieces of K-tile 1 are therefore in flight during the epilogue); step KT-1 ends
without fragment reads (the next statement starts with them).
"""
import argparse

ap = argparse.ArgumentParser()
ap.add_argument("--sched", default="3,3,3,3,2,2,0,0")
args = ap.parse_args()
sched = [int(x) for x in args.sched.split(",")]
assert len(sched) == 8 and sum(sched) == 16 and max(sched) <= 4
k_first = sum(sched[:4])


def acc(m, n):
    b = 4 * (8 * m + n)
    return f"a[{b}:{b + 3}]"


def frag(buf, kind, i):  # kind 0 = A, 1 = W
    b = 64 * buf + 32 * kind + 4 * i
    return f"v[{b}:{b + 3}]"


def piece(c):
    const = (0 if c < 8 else 32768) + (c & 7) * 4096
    return [f"s_add_u32 m0, s25, {const}",
            "s_nop 0",  # an SALU write of M0 needs one wait state before the LDS-DMA that reads it (hipcc inserts the same s_nop for its builtin)
            f"buffer_load_dwordx4 v{132 + c}, {'%[srda]' if c < 8 else '%[srdw]'}, {'s20' if c < 8 else 's21'} offen lds"]


def phase(emit, h, zero_c=False, reads=True):
    hn = h ^ 1
    rdA, rdW = (128, 129) if hn == 0 else (130, 131)
    rd = []
    if reads:
        for i in range(8):
            rd.append(f"ds_read_b128 {frag(hn, 1, i)}, v{rdW} offset:{i * 2048}")
        for i in range(8):
            rd.append(f"ds_read_b128 {frag(hn, 0, i)}, v{rdA} offset:{i * 2048}")
    pcs = sched[0:4] if h == 1 else sched[4:8]
    base = 0 if h == 1 else k_first
    slots = [[] for _ in range(64)]
    for i, r in enumerate(rd):
        slots[i].append(r)
    c = base
    for j in range(4):
        for p in range(pcs[j]):
            slots[16 * j + 9 + 2 * p].extend(piece(c))
            c += 1
    k = 0
    for j in range(4):
        mg, ng = j >> 1, j & 1
        for i in range(16):
            m_, n_ = 4 * mg + (i >> 2), 4 * ng + (i & 3)
            emit(f"v_mfma_f32_16x16x32_f16 {acc(m_, n_)}, {frag(h, 1, n_)}, {frag(h, 0, m_)}, {'0' if zero_c else acc(m_, n_)}")
            for s in slots[k]:
                emit(s)
            k += 1


def step(emit, zero_c=False, switch=False, last=False):
    phase(emit, 0, zero_c=zero_c)
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    emit("s_barrier")
    if switch:  # the loader leaves this tile: K-tile 0 of the next one
        emit("s_sub_u32 s20, %[nexta], %[soffa]")
        for c in range(8):
            emit(f"v_add_u32 v{132 + c}, s20, v{132 + c}")
        emit("s_mov_b32 s20, 0")
        emit("s_mov_b32 s21, %[nextw]")
    else:
        emit("s_add_u32 s20, s20, 128")
        emit("s_add_u32 s21, s21, 128")
    emit("s_xor_b32 s25, s25, 0x10000")
    emit("v_xor_b32 v128, 0x10000, v128")
    emit("v_xor_b32 v129, 0x10000, v129")
    phase(emit, 1, reads=not last)
    emit("v_xor_b32 v130, 0x10000, v130")
    emit("v_xor_b32 v131, 0x10000, v131")


def setup(emit):
    emit("v_mov_b32 v128, %[rda]")
    emit("v_mov_b32 v129, %[rdw]")
    emit("v_xor_b32 v130, 64, v128")  # k-half 1 = 16-byte chunk index ^ 4
    emit("v_xor_b32 v131, 64, v129")
    # the descriptor's bounds check covers the VGPR offset only: the tile's ROWS of A go into it (rows past M then read as zeros wherever the
    # scalar offset stands), the scalar offset s20 carries the K position alone; W rows are always in range, its tile base stays scalar
    emit("v_add_u32 v132, %[soffa], %[offa0]")
    for c in range(1, 8):
        emit(f"v_add_u32 v{132 + c}, %[rsa], v{131 + c}")
    emit("v_mov_b32 v140, %[offw0]")
    for c in range(9, 16):
        emit(f"v_add_u32 v{132 + c}, %[rsw], v{131 + c}")


def as_c_string(name, lines):
    out = [f"#define {name} \\"]
    for line in lines:
        out.append(f'    "{line}\\n\\t" \\')
    out.append('    ""')
    return "\n".join(out)


# ---- prologue of a workgroup's first tile: K-tile 0 completely + the first pieces of K-tile 1, then wait for K-tile 0 and meet
pro = []
setup(pro.append)
pro.append("s_mov_b32 s20, 0")
pro.append("s_mov_b32 s21, %[soffw]")
pro.append("s_mov_b32 s25, %[dst0]")
for c in range(16):
    pro.extend(piece(c))
pro.append("s_add_u32 s20, s20, 128")
pro.append("s_add_u32 s21, s21, 128")
pro.append("s_xor_b32 s25, s25, 0x10000")
for c in range(k_first):
    pro.extend(piece(c))
pro.append(f"s_waitcnt vmcnt({k_first})")
pro.append("s_barrier")

# ---- the K-loop of one tile. Entry state: K-tile 0 landed (every wave past the barrier), kFirst pieces of K-tile 1 in flight.
body = []
e = body.append
setup(e)
e("s_mov_b32 s20, 128")
e("s_add_u32 s21, %[soffw], 128")
e("s_xor_b32 s25, %[dst0], 0x10000")
e("s_mov_b32 s24, %[iters]")
for i in range(8):
    e(f"ds_read_b128 {frag(0, 1, i)}, v129 offset:{i * 2048}")
for i in range(8):
    e(f"ds_read_b128 {frag(0, 0, i)}, v128 offset:{i * 2048}")
e("s_waitcnt lgkmcnt(0)")
step(e, zero_c=True)
e("1:")
e("s_waitcnt lgkmcnt(0)")
step(e)
e("s_sub_u32 s24, s24, 1")
e("s_cmp_lg_u32 s24, 0")
e("s_cbranch_scc1 1b")
e("s_waitcnt lgkmcnt(0)")
step(e, switch=True)
e("s_waitcnt lgkmcnt(0)")
step(e, last=True)
e("s_nop 15")  # the epilogue reads the accumulators with v_accvgpr_read: nobody inserts the MFMA -> VALU wait states for asm
e("s_nop 15")

print("// GENERATED by scripts/gen_gemm_quad_asm.py --sched " + args.sched + " -- do not edit; see that script for the register map.")
print(f"#define MDR_QUAD_KFIRST {k_first}")
print(as_c_string("MDR_QUAD_PROLOGUE_ASM", pro))
print(as_c_string("MDR_QUAD_KLOOP_ASM", body))
regs = [f'"v{i}"' for i in range(148)]
print("#define MDR_QUAD_CLOBBER_V " + ", ".join(regs))
print("#define MDR_QUAD_CLOBBER_A " + ", ".join(f'"a{i}"' for i in range(256)))
