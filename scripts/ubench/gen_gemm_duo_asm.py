#!/usr/bin/env python3
"""ARCHIVED EXPERIMENT (round 3, see mdr_encoder_gemm_duo.inl.txt beside this file). Generates mdr_encoder_gemm_duo_loop.inc: the hand-scheduled K-loops of gemm_duo_kernel -- a persistent
256x128x64 f16 GEMM on four waves of 128x64 (one wave per SIMD), a THREE-slot LDS ring (48 KiB slots, a K-tile of extra DMA lead) and TWO
accumulator sets, so that the epilogue of tile j-1 is woven into the K-loop of tile j (one wave per SIMD cannot hide an epilogue any other way;
measured alone it costs 30-45 % of a K = 768 tile).

    python scripts/gen_gemm_duo_asm.py > multihop_dense_retrieval_amd/csrc/mdr_encoder_gemm_duo_loop.inc

Register map (per wave; all of it on the statements' clobber lists):
  a[0:127]   accumulator set 0, a[128:255] set 1; tile (m, n) of set S at a[128 S + 4 (4 m + n)] -- live ACROSS statements
  v[0:31] / v[32:47]    A / W fragments of k-half 0;  v[48:79] / v[80:95] of k-half 1
  v[96+2s] / v[97+2s]   LDS read address of A / W in slot s, k-half 0;  v[102+2s] / v[103+2s] k-half 1
  v[108:119] buffer offsets of the wave's 12 DMA pieces (8 of A, 4 of W)
  v[120:135] bias of the wave's 64 columns (4 per n-fragment), v[136:151] a unit's values, v[152:159] packed halves
  v[160:163] scratch write addresses, v164 scratch read address, v[166:181] store data, v182 / v183 store offsets of rows 0-7 / 8-15, v184 bias offset
  s20 / s21 buffer offset of the loader's K-tile in A / W; s24 loop counter; s25 = wave * 1024; s26 output offset of the unit being stored
One step = one K-tile in slot T % 3: phase 0 (32 MFMAs on k-half 0; reads k-half 1 of the slot; the rest of K-tile T+2's DMA pieces) |
s_waitcnt vmcnt(N) lgkmcnt(0) + s_barrier (K-tile T+1 landed, slot free; the 12 pieces of T+2 stay in flight) | phase 1 (32 MFMAs on k-half 1;
reads k-half 0 of K-tile T+1; the first pieces of K-tile T+3 into the freed slot). Epilogue unit u (16 rows x 128 B of output of the PREVIOUS
tile) rides along: values read / biased / converted in phase 1 of step u+1, through the per-wave LDS scratch in phase 0 of step u+2, stored
(buffer_store: rows past M are dropped by the descriptor) in phase 1 of step u+2.
"""
import argparse

ap = argparse.ArgumentParser()
ap.add_argument("--sched", default="4,4,4,0", help="DMA pieces per 16-MFMA sub-phase: 2 of phase 1, then 2 of the next phase 0")
ap.add_argument("--abl", type=int, default=0, help="measurement builds (results wrong): 1 = no epilogue ops, 2 = arithmetic only, 3 = arithmetic + LDS (no stores)")
args = ap.parse_args()
sched = [int(x) for x in args.sched.split(",")]
assert len(sched) == 4 and sum(sched) == 12 and max(sched) <= 4
K_FIRST = sched[0] + sched[1]
SLOT = 49152
SCRATCH = 3 * SLOT  # + wave * 2048, added by the host code in the addresses it passes

F16, F32 = "f16", "f32"


def acc(S, m, n):
    b = 128 * S + 4 * (4 * m + n)
    return f"a[{b}:{b + 3}]"


def frag(buf, kind, i):  # kind 0 = A, 1 = W
    b = (0, 32, 48, 80)[2 * buf + kind] + 4 * i
    return f"v[{b}:{b + 3}]"


def piece(c, slot):
    const = slot * SLOT + (0 if c < 8 else 32768) + (c & 7) * 4096
    return [f"s_add_u32 m0, s25, {const}",
            "s_nop 0",
            f"buffer_load_dwordx4 v{108 + c}, {'%[srda]' if c < 8 else '%[srdw]'}, {'s20' if c < 8 else 's21'} offen lds"]


# ---- epilogue unit u of the previous tile (accumulator set S): three op lists
def unit_ops(kind, S, u):
    """returns (compute, lds, stores): instruction lists. f16: unit = m-fragment u (4 n-fragments, 16 rows x 64 columns x 2 B);
    f32: unit = (m-fragment u >> 1, column half u & 1) (2 n-fragments, 16 rows x 32 columns x 4 B)."""
    comp, lds, st = [], [], []
    if kind == F16:
        mt = u
        for q in range(4):
            b = 128 * S + 4 * (4 * mt + q)
            for r in range(4):
                comp.append(f"v_accvgpr_read_b32 v{136 + 4 * q + r}, a{b + r}")
            comp.append(f"v_pk_add_f32 v[{136 + 4 * q}:{137 + 4 * q}], v[{136 + 4 * q}:{137 + 4 * q}], v[{120 + 4 * q}:{121 + 4 * q}]")
            comp.append(f"v_pk_add_f32 v[{138 + 4 * q}:{139 + 4 * q}], v[{138 + 4 * q}:{139 + 4 * q}], v[{122 + 4 * q}:{123 + 4 * q}]")
            comp.append(f"v_cvt_pk_f16_f32 v{152 + 2 * q}, v{136 + 4 * q}, v{137 + 4 * q}")
            comp.append(f"v_cvt_pk_f16_f32 v{153 + 2 * q}, v{138 + 4 * q}, v{139 + 4 * q}")
            lds.append(f"ds_write_b64 v{160 + q}, v[{152 + 2 * q}:{153 + 2 * q}]")
        lds.append("ds_read_b128 v[166:169], v164")
        lds.append("ds_read_b128 v[170:173], v164 offset:1024")
        st.append("buffer_store_dwordx4 v[166:169], v182, %[srdo], s26 offen")
        st.append("buffer_store_dwordx4 v[170:173], v183, %[srdo], s26 offen")
        st.append("s_add_u32 s26, s26, %[ldo16]")  # next unit: 16 rows down
    else:
        mt, hf = u >> 1, u & 1
        base_t = 136 + 8 * hf  # the two units of a step use separate value registers
        for q2 in range(2):
            q = 2 * hf + q2
            b = 128 * S + 4 * (4 * mt + q)
            t = base_t + 4 * q2
            for r in range(4):
                comp.append(f"v_accvgpr_read_b32 v{t + r}, a{b + r}")
            comp.append(f"v_pk_add_f32 v[{t}:{t + 1}], v[{t}:{t + 1}], v[{120 + 4 * q}:{121 + 4 * q}]")
            comp.append(f"v_pk_add_f32 v[{t + 2}:{t + 3}], v[{t + 2}:{t + 3}], v[{122 + 4 * q}:{123 + 4 * q}]")
            lds.append(f"ds_write_b128 v{160 + q2}, v[{t}:{t + 3}]")
        d = 166 + 8 * hf
        lds.append(f"ds_read_b128 v[{d}:{d + 3}], v164")
        lds.append(f"ds_read_b128 v[{d + 4}:{d + 7}], v164 offset:1024")
        st.append(f"buffer_store_dwordx4 v[{d}:{d + 3}], v182, %[srdo], s26 offen{' offset:128' if hf else ''}")
        st.append(f"buffer_store_dwordx4 v[{d + 4}:{d + 7}], v183, %[srdo], s26 offen{' offset:128' if hf else ''}")
        if hf:
            st.append("s_add_u32 s26, s26, %[ldo16]")
    return comp, lds, st


def n_units(kind):
    return 8 if kind == F16 else 16


def units_per_step(kind):
    return 1 if kind == F16 else 2


def spread(ops, n_slots, first=0, last=None):
    """distribute ops evenly over slots [first, last)"""
    last = n_slots if last is None else last
    out = [[] for _ in range(n_slots)]
    span = last - first
    for i, op in enumerate(ops):
        out[first + (i * span) // max(len(ops), 1)].append(op)
    return out


def phase(emit, h, S, rd_slot, ld_slot, zero_c, reads, extra):
    """32 MFMAs into accumulator set S from fragment buffer h; behind MFMA i: reads (first 12 slots), DMA pieces per sched, extra[i]"""
    hn = h ^ 1
    rdA = (96 if hn == 0 else 102) + 2 * rd_slot
    rd = []
    if reads:
        rd = [f"ds_read_b128 {frag(hn, 1, i)}, v{rdA + 1} offset:{i * 2048}" for i in range(4)] + \
             [f"ds_read_b128 {frag(hn, 0, i)}, v{rdA} offset:{i * 2048}" for i in range(8)]
    pcs = sched[0:2] if h == 1 else sched[2:4]
    c = 0 if h == 1 else K_FIRST
    slots = [[] for _ in range(32)]
    for i, r in enumerate(rd):
        slots[i].append(r)
    for j in range(2):
        for p in range(pcs[j]):
            slots[16 * j + 9 + 2 * p].extend(piece(c, ld_slot))
            c += 1
    k = 0
    for j in range(2):
        for i in range(16):
            m_, n_ = 4 * j + (i >> 2), i & 3
            emit(f"v_mfma_f32_16x16x32_f16 {acc(S, m_, n_)}, {frag(h, 1, n_)}, {frag(h, 0, m_)}, {'0' if zero_c else acc(S, m_, n_)}")
            for s in slots[k]:
                emit(s)
            for s in extra[k]:
                emit(s)
            k += 1


def step(emit, T, S, kind=None, switch=False, last=False, first=False, e_prev=0):
    """K-tile T of a tile (slot T % 3). kind: weave the epilogue of accumulator set S ^ 1. Returns the number of epilogue VMEM ops issued in
    phase 1 (the next barrier has to leave them in flight)."""
    s = T % 3
    none = [[] for _ in range(32)]
    ex0, ex1, vm = none, none, 0
    if kind is not None:
        ups = units_per_step(kind)
        comp1, lds0, st1 = [], [], []
        for k in range(ups):
            uc = (T - 1) * ups + k  # unit computed in phase 1 of this step
            ul = (T - 2) * ups + k  # unit staged through LDS in phase 0 and stored in phase 1 of this step
            if 0 <= uc < n_units(kind):
                comp1 += unit_ops(kind, S ^ 1, uc)[0]
            if 0 <= ul < n_units(kind):
                _, l, t = unit_ops(kind, S ^ 1, ul)
                lds0 += l
                st1 += t
        if args.abl == 1:
            comp1, lds0, st1 = [], [], []
        elif args.abl == 2:
            lds0, st1 = [], [x for x in st1 if not x.startswith("buffer_store")]
        elif args.abl == 3:
            st1 = [x for x in st1 if not x.startswith("buffer_store")]
        ex0 = spread(lds0, 32, 12, 18)              # behind the fragment reads, >= 14 MFMAs ahead of the barrier's lgkmcnt(0)
        ex1 = spread(st1, 32, 0, 8)                 # stores first (their data arrived before the barrier) ...
        c1 = spread(comp1, 32, 2, 32)               # ... then the next unit's arithmetic
        ex1 = [a + b for a, b in zip(ex1, c1)]
        vm = sum(1 for x in st1 if x.startswith("buffer_store"))
    emit("s_waitcnt lgkmcnt(0)")
    phase(emit, 0, S, s, (s + 2) % 3, first, True, ex0)
    emit(f"s_waitcnt vmcnt({12 + e_prev}) lgkmcnt(0)")
    emit("s_barrier")
    if switch:
        emit("s_mov_b32 s20, %[nexta]")
        emit("s_mov_b32 s21, %[nextw]")
    else:
        emit("s_add_u32 s20, s20, 128")
        emit("s_add_u32 s21, s21, 128")
    phase(emit, 1, S, (s + 1) % 3, s, False, not last, ex1)
    return vm


def setup(emit, loader_k):
    for s in range(3):
        emit(f"v_add_u32 v{96 + 2 * s}, {s * SLOT}, %[rda]")
        emit(f"v_add_u32 v{97 + 2 * s}, {s * SLOT}, %[rdw]")
        emit(f"v_xor_b32 v{102 + 2 * s}, 64, v{96 + 2 * s}")
        emit(f"v_xor_b32 v{103 + 2 * s}, 64, v{97 + 2 * s}")
    emit("v_mov_b32 v108, %[offa0]")
    for c in range(1, 8):
        emit(f"v_add_u32 v{108 + c}, %[rsa], v{107 + c}")
    emit("v_mov_b32 v116, %[offw0]")
    for c in range(9, 12):
        emit(f"v_add_u32 v{108 + c}, %[rsw], v{107 + c}")
    emit("s_mov_b32 s25, %[dst0]")
    emit(f"s_add_u32 s20, %[soffa], {128 * loader_k}")
    emit(f"s_add_u32 s21, %[soffw], {128 * loader_k}")


def entry_reads(emit):
    for i in range(4):
        emit(f"ds_read_b128 {frag(0, 1, i)}, v97 offset:{i * 2048}")
    for i in range(8):
        emit(f"ds_read_b128 {frag(0, 0, i)}, v96 offset:{i * 2048}")


def as_c_string(name, lines):
    out = [f"#define {name} \\"]
    for line in lines:
        out.append(f'    "{line}\\n\\t" \\')
    out.append('    ""')
    return "\n".join(out)


# ---- prologue of a workgroup's first tile: K-tiles 0 and 1 completely, the first pieces of K-tile 2; wait for K-tile 0, meet
pro = []
setup(pro.append, 0)
for c in range(12):
    pro.extend(piece(c, 0))
pro.append("s_add_u32 s20, s20, 128")
pro.append("s_add_u32 s21, s21, 128")
for c in range(12):
    pro.extend(piece(c, 1))
pro.append("s_add_u32 s20, s20, 128")
pro.append("s_add_u32 s21, s21, 128")
for c in range(K_FIRST):
    pro.extend(piece(c, 2))
pro.append(f"s_waitcnt vmcnt({12 + K_FIRST})")
pro.append("s_barrier")


def tile_block(kind, S, long_k):
    """the K-loop of one tile into accumulator set S. kind None: no epilogue (a workgroup's first tile; KT = 3 (iters + 1), rolled).
    Otherwise the first 12 steps are unrolled with the previous tile's epilogue woven in; long_k: a rolled part and three peeled steps follow."""
    b = []
    e = b.append
    setup(e, 2)
    if kind is not None:
        e("s_mov_b32 s26, %[sout]")
        if kind == F16:
            for q in range(4):
                e(f"v_xor_b32 v{160 + q}, {32 * q}, %[vwr]")
        else:
            e("v_mov_b32 v160, %[vwr]")
            e("v_xor_b32 v161, 64, %[vwr]")
        e("v_mov_b32 v164, %[vrd]")
        e("v_mov_b32 v182, %[vst0]")
        e("v_mov_b32 v183, %[vst1]")
        for q in range(4):  # bias of the PREVIOUS tile's columns
            e(f"buffer_load_dwordx4 v[{120 + 4 * q}:{123 + 4 * q}], %[vboff], %[srdb], %[sbias] offen offset:{64 * q}")
    entry_reads(e)
    n_bias = 4 if kind is not None else 0
    if kind is None:
        # step 0 (C = 0), then a rolled loop of 3 steps, then the last three steps with the loader moving to the next tile; KT = 3 iters + 6... keep it simple:
        # steps 0,1,2 peeled (first), `iters` x 3 rolled, steps KT-3..KT-1 peeled -> KT = 6 + 3 iters (iters may be 0 is not supported: KT >= 9)
        ev = step(e, 0, S, first=True, e_prev=n_bias)
        step(e, 1, S)
        step(e, 2, S)
        e("s_mov_b32 s24, %[iters]")
        e("1:")
        step(e, 0, S)
        step(e, 1, S)
        step(e, 2, S)
        e("s_sub_u32 s24, s24, 1")
        e("s_cmp_lg_u32 s24, 0")
        e("s_cbranch_scc1 1b")
        step(e, 0, S, switch=True)
        step(e, 1, S)
        step(e, 2, S, last=True)
    else:
        ev = n_bias
        for T in range(12):
            sw = (not long_k) and T == 9
            ev = step(e, T, S, kind=kind, first=(T == 0), switch=sw, last=(not long_k and T == 11), e_prev=ev)
        if long_k:
            e("s_mov_b32 s24, %[iters]")
            e("1:")
            step(e, 0, S)
            step(e, 1, S)
            step(e, 2, S)
            e("s_sub_u32 s24, s24, 1")
            e("s_cmp_lg_u32 s24, 0")
            e("s_cbranch_scc1 1b")
            step(e, 0, S, switch=True)
            step(e, 1, S)
            step(e, 2, S, last=True)
    e("s_nop 15")
    e("s_nop 15")
    return b


print("// GENERATED by scripts/gen_gemm_duo_asm.py --sched " + args.sched + " -- do not edit; see that script for the register map.")
print(f"#define MDR_DUO_KFIRST {K_FIRST}")
print(as_c_string("MDR_DUO_PRO_ASM", pro))
print(as_c_string("MDR_DUO_PLAIN_ASM", tile_block(None, 0, False)))
for kind in (F16, F32):
    for S in (0, 1):
        print(as_c_string(f"MDR_DUO_K12_{kind.upper()}_S{S}_ASM", tile_block(kind, S, False)))
for S in (0, 1):
    print(as_c_string(f"MDR_DUO_KL_F32_S{S}_ASM", tile_block(F32, S, True)))
print("#define MDR_DUO_CLOBBER_V " + ", ".join(f'"v{i}"' for i in range(185)))
print("#define MDR_DUO_CLOBBER_A " + ", ".join(f'"a{i}"' for i in range(256)))
