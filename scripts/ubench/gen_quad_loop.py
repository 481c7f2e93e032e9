#!/usr/bin/env python3
"""Generates the K-loop of a 256x256x64 f16 GEMM tile for FOUR waves of 128x128 (one wave per SIMD) as one hand-scheduled asm block with
fixed registers, for scripts/ubench/quad_loop.hip -- the experiment behind DESIGN.md section 4 "four-wave K-loop".

    python scripts/ubench/gen_quad_loop.py [--sched a,b,c,d,e,f,g,h] [--reads-per-slot 1] > scripts/ubench/quad_loop.inc

Register map (per wave):
  a[0:255]   accumulators, tile (m, n) at a[4 (8 m + n)]
  v[0:31]    A fragments buffer 0 (k-half 0), v[32:63] W buffer 0, v[64:95] A buffer 1 (k-half 1), v[96:127] W buffer 1
  v128/v129  LDS read addresses of A / W for k-half 0 (slot included), v130/v131 for k-half 1
  v[132:147] DMA source offsets of the wave's 16 pieces (bytes from the running A / W base)
  s[20:21]   A base of the loader's K-tile, s[22:23] W base, s24 K-tiles left, s25 LDS destination of piece 0 (wave and slot included)
One K-tile = phase 0 (k-half 0: 64 MFMAs on buffer 0, reads k-half 1 of the same slot into buffer 1) | s_waitcnt + s_barrier |
phase 1 (64 MFMAs on buffer 1, reads k-half 0 of the NEXT K-tile from the other slot into buffer 0). Behind every MFMA one slot for
another instruction: the 16 reads of a phase behind its first MFMAs, the DMA pieces (s_add m0 + global_load_lds) per `sched`.
"""
import argparse

ap = argparse.ArgumentParser()
ap.add_argument("--sched", default="3,3,3,3,2,2,0,0", help="DMA pieces per sub-phase (16 MFMAs): 4 of phase 1, then 4 of the next phase 0")
ap.add_argument("--nf", type=int, default=8, help="n-fragments per wave: 8 = 128x128 wave tile (256x256 workgroup tile), 4 = 128x64 (256x128)")
ap.add_argument("--stores", type=int, default=0, help="global_store_dwordx4 per K-tile and wave behind the barrier (epilogue traffic of an overlapped design)")
ap.add_argument("--no-mfma", action="store_true")
ap.add_argument("--no-reads", action="store_true")
ap.add_argument("--no-dma", action="store_true")
ap.add_argument("--no-barrier", action="store_true")
args = ap.parse_args()
sched = [int(x) for x in args.sched.split(",")]
NF = args.nf
SUBS = NF // 4 * 2          # 16-MFMA sub-phases per phase
NPIECE = 8 + NF             # DMA pieces per wave and K-tile (A: 8 x 32 rows, W: NF x 32 rows)
assert len(sched) == 2 * SUBS and sum(sched) == NPIECE and max(sched) <= 4
k_first = sum(sched[:SUBS])
A0, W0, A1, W1 = 0, 32, 32 + 4 * NF, 64 + 4 * NF   # fragment register bases

out = []
emit = out.append


def acc(m, n):
    b = 4 * (NF * m + n)
    return f"a[{b}:{b + 3}]"


def frag(buf, kind, i):  # kind 0 = A, 1 = W
    b = (A0, W0, A1, W1)[2 * buf + kind] + 4 * i
    return f"v[{b}:{b + 3}]"


def piece(c):
    const = (0 if c < 8 else 32768) + (c & 7) * 4096
    if args.no_dma:
        return []
    return [f"s_add_u32 m0, s25, {const}", "s_nop 0", f"global_load_lds_dwordx4 v{132 + c}, {'s[20:21]' if c < 8 else 's[22:23]'}"]


def phase(h):
    hn = h ^ 1
    rdA, rdW = (128, 129) if hn == 0 else (130, 131)
    reads = []
    for i in range(NF):  # W first (used by every MFMA row), then A
        reads.append(f"ds_read_b128 {frag(hn, 1, i)}, v{rdW} offset:{i * 2048}")
    for i in range(8):
        reads.append(f"ds_read_b128 {frag(hn, 0, i)}, v{rdA} offset:{i * 2048}")
    if args.no_reads:
        reads = []
    pcs = sched[0:SUBS] if h == 1 else sched[SUBS:2 * SUBS]
    base = 0 if h == 1 else k_first
    # slots: index 0..16*SUBS-1 behind MFMA i
    slots = [[] for _ in range(16 * SUBS)]
    for i, r in enumerate(reads):
        slots[i].append(r)
    c = base
    for j in range(SUBS):
        for p in range(pcs[j]):
            slots[16 * j + 9 + 2 * p].extend(piece(c))
            c += 1
    if h == 1:
        for p in range(args.stores):
            slots[2 + 3 * p].append(f"global_store_dwordx4 v148, v[152:155], s[30:31] offset:{256 * p}")
    k = 0
    for j in range(SUBS):
        mg, ng = (j >> 1, j & 1) if NF == 8 else (j, 0)
        for i in range(16):
            q, n = i >> 2, i & 3
            m_, n_ = 4 * mg + q, 4 * ng + n
            if not args.no_mfma:
                emit(f"v_mfma_f32_16x16x32_f16 {acc(m_, n_)}, {frag(h, 1, n_)}, {frag(h, 0, m_)}, {acc(m_, n_)}")
            for s in slots[k]:
                emit(s)
            k += 1


# ---- prologue (inputs: %[A] s64, %[W] s64, %[kt] s32, %[rda] v (lane part of the A read address, k-half 0), %[rdw] v, %[sw] v (xor that turns a
# k-half-0 address into k-half 1), %[off0] v (DMA offset of piece 0), %[rs32] s (32 rows x row stride, bytes), %[dst0] s (LDS dst of piece 0, slot 0)
emit("s_mov_b64 s[20:21], %[A]")
emit("s_mov_b64 s[22:23], %[W]")
emit("s_mov_b32 s24, %[kt]")
emit("s_mov_b32 s25, %[dst0]")
emit("v_mov_b32 v128, %[rda]")
emit("v_mov_b32 v129, %[rdw]")
emit("v_xor_b32 v130, %[sw], %[rda]")
emit("v_xor_b32 v131, %[sw], %[rdw]")
emit("v_mov_b32 v132, %[off0]")
for c in range(1, 8):
    emit(f"v_add_u32 v{132 + c}, %[rs32], v{131 + c}")
emit("v_mov_b32 v140, %[off0]")
for c in range(9, NPIECE):
    emit(f"v_add_u32 v{132 + c}, %[rs32], v{131 + c}")
emit("s_mov_b64 s[30:31], %[st]")
emit("v_mov_b32 v148, %[stoff]")
for i in range(256):
    emit(f"v_accvgpr_write_b32 a{i}, 0")
emit("s_memtime s[26:27]")
for c in range(NPIECE):
    for s in piece(c):
        emit(s)
emit("s_add_u32 s20, s20, 128")
emit("s_addc_u32 s21, s21, 0")
emit("s_add_u32 s22, s22, 128")
emit("s_addc_u32 s23, s23, 0")
emit("s_xor_b32 s25, s25, 0x10000")
for c in range(k_first):
    for s in piece(c):
        emit(s)
emit(f"s_waitcnt vmcnt({0 if args.no_dma else k_first})")
if not args.no_barrier:
    emit("s_barrier")
if not args.no_reads:
    for i in range(NF):
        emit(f"ds_read_b128 {frag(0, 1, i)}, v129 offset:{i * 2048}")
    for i in range(8):
        emit(f"ds_read_b128 {frag(0, 0, i)}, v128 offset:{i * 2048}")
emit("1:")
emit("s_waitcnt lgkmcnt(0)")
phase(0)
emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
if not args.no_barrier:
    emit("s_barrier")
# the loader moves on to K-tile T+2 (slot just freed); k-half-0 reads move to the other slot
emit("s_add_u32 s20, s20, 128")
emit("s_addc_u32 s21, s21, 0")
emit("s_add_u32 s22, s22, 128")
emit("s_addc_u32 s23, s23, 0")
emit("s_xor_b32 s25, s25, 0x10000")
emit("v_xor_b32 v128, 0x10000, v128")
emit("v_xor_b32 v129, 0x10000, v129")
phase(1)
emit("v_xor_b32 v130, 0x10000, v130")
emit("v_xor_b32 v131, 0x10000, v131")
if args.stores:
    emit("s_add_u32 s30, s30, 0x10000")
    emit("s_addc_u32 s31, s31, 0")
emit("s_sub_u32 s24, s24, 1")
emit("s_cmp_lg_u32 s24, 0")
emit("s_cbranch_scc1 1b")
emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
emit("s_nop 15")
emit("s_nop 15")
emit("s_memtime s[28:29]")
emit("v_accvgpr_read_b32 %[o0], a0")
emit(f"v_accvgpr_read_b32 %[o1], a{32 * NF - 1}")
emit("s_waitcnt lgkmcnt(0)")
emit("s_sub_u32 s26, s28, s26")
emit("s_subb_u32 s27, s29, s27")
emit("v_mov_b32 %[t0], s26")
emit("v_mov_b32 %[t1], s27")

print("// generated by scripts/ubench/gen_quad_loop.py " + " ".join(f"--{k.replace('_', '-')}={v}" if not isinstance(v, bool) else (f"--{k.replace('_', '-')}" if v else "") for k, v in vars(args).items()))
for line in out:
    print(f'"{line}\\n\\t"')
