#!/usr/bin/env python3
"""K-loop skeleton of a 256x128x64 f16 GEMM tile on four waves of 128x64 with a THREE-slot LDS ring (48 KiB slots) for
scripts/ubench/quad_loop.hip (-DQUAD_NF=4): does a K-tile of extra DMA lead (pieces of K-tile T+2 still in flight across the barrier of
step T) lift the L2->LDS rate that the two-slot loops are held to (~11 TB/s over the chip)?  Loop unrolled by 3 (static slots).
  a[0:127] accumulators; v[0:31] A / v[32:47] W fragments of k-half 0, v[48:79] / v[80:95] of k-half 1
  v[100+2s], v[101+2s]: LDS read address of A / W in slot s for k-half 0, v[106+2s], v[107+2s] for k-half 1; v[132:143] DMA offsets
  s[20:21] / s[22:23] A / W base of the loader's K-tile, s24 loop counter, s25 = wave * 1024 (LDS destination of piece 0 in slot 0)
"""
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--sched", default="4,4,4,0")
ap.add_argument("--stores", type=int, default=0)
ap.add_argument("--no-dma", action="store_true")
args = ap.parse_args()
sched = [int(x) for x in args.sched.split(",")]
assert len(sched) == 4 and sum(sched) == 12 and max(sched) <= 4
k_first = sched[0] + sched[1]
SLOT = 49152
out = []
emit = out.append

def acc(m, n):
    b = 4 * (4 * m + n); return f"a[{b}:{b + 3}]"
def frag(buf, kind, i):
    b = (0, 32, 48, 80)[2 * buf + kind] + 4 * i; return f"v[{b}:{b + 3}]"
def piece(c, slot):
    const = slot * SLOT + (0 if c < 8 else 32768) + (c & 7) * 4096
    if args.no_dma: return []
    return [f"s_add_u32 m0, s25, {const}", "s_nop 0", f"global_load_lds_dwordx4 v{132 + c}, {'s[20:21]' if c < 8 else 's[22:23]'}"]

def phase(h, rd_slot, ld_slot):
    hn = h ^ 1
    rdA = (100 if hn == 0 else 106) + 2 * rd_slot
    rdW = rdA + 1
    reads = [f"ds_read_b128 {frag(hn, 1, i)}, v{rdW} offset:{i * 2048}" for i in range(4)] + [f"ds_read_b128 {frag(hn, 0, i)}, v{rdA} offset:{i * 2048}" for i in range(8)]
    pcs = sched[0:2] if h == 1 else sched[2:4]
    base = 0 if h == 1 else k_first
    slots = [[] for _ in range(32)]
    for i, r in enumerate(reads): slots[i].append(r)
    c = base
    for j in range(2):
        for p in range(pcs[j]):
            slots[16 * j + 9 + 2 * p].extend(piece(c, ld_slot)); c += 1
    if h == 1:
        for p in range(args.stores):
            slots[2 + 3 * p].append(f"global_store_dwordx4 v148, v[152:155], s[30:31] offset:{256 * p}")
    k = 0
    for j in range(2):
        for i in range(16):
            m_, n_ = 4 * j + (i >> 2), i & 3
            emit(f"v_mfma_f32_16x16x32_f16 {acc(m_, n_)}, {frag(h, 1, n_)}, {frag(h, 0, m_)}, {acc(m_, n_)}")
            for s in slots[k]: emit(s)
            k += 1

def advance():
    emit("s_add_u32 s20, s20, 128"); emit("s_addc_u32 s21, s21, 0"); emit("s_add_u32 s22, s22, 128"); emit("s_addc_u32 s23, s23, 0")

def step(s):  # K-tile in slot s; loader: rest of K-tile T+2 (slot (s+2)%3) in phase 0, start of K-tile T+3 (slot s) in phase 1
    emit("s_waitcnt lgkmcnt(0)")
    phase(0, s, (s + 2) % 3)
    # K-tile T+1 landed (the 12 pieces of T+2 may still fly), slot s read completely
    emit(f"s_waitcnt vmcnt({0 if args.no_dma else 12 + args.stores}) lgkmcnt(0)")
    emit("s_barrier")
    advance()
    phase(1, (s + 1) % 3, s)
    if args.stores:
        emit("s_add_u32 s30, s30, 0x10000"); emit("s_addc_u32 s31, s31, 0")

emit("s_mov_b64 s[20:21], %[A]"); emit("s_mov_b64 s[22:23], %[W]"); emit("s_mov_b32 s24, %[kt]"); emit("s_mov_b32 s25, %[dst0]")
for s in range(3):
    emit(f"v_add_u32 v{100 + 2 * s}, {s * SLOT}, %[rda]"); emit(f"v_add_u32 v{101 + 2 * s}, {s * SLOT}, %[rdw]")
    emit(f"v_xor_b32 v{106 + 2 * s}, 64, v{100 + 2 * s}"); emit(f"v_xor_b32 v{107 + 2 * s}, 64, v{101 + 2 * s}")
emit("v_mov_b32 v132, %[off0]")
for c in range(1, 8): emit(f"v_add_u32 v{132 + c}, %[rs32], v{131 + c}")
emit("v_mov_b32 v140, %[off0]")
for c in range(9, 12): emit(f"v_add_u32 v{132 + c}, %[rs32], v{131 + c}")
emit("s_mov_b64 s[30:31], %[st]"); emit("v_mov_b32 v148, %[stoff]")
for i in range(128): emit(f"v_accvgpr_write_b32 a{i}, 0")
emit("s_memtime s[26:27]")
# prologue: K-tiles 0 and 1 completely, the first pieces of K-tile 2
for c in range(12):
    for x in piece(c, 0): emit(x)
advance()
for c in range(12):
    for x in piece(c, 1): emit(x)
advance()
for c in range(k_first):
    for x in piece(c, 2): emit(x)
emit(f"s_waitcnt vmcnt({0 if args.no_dma else 12 + k_first})")
emit("s_barrier")
for i in range(4): emit(f"ds_read_b128 {frag(0, 1, i)}, v101 offset:{i * 2048}")
for i in range(8): emit(f"ds_read_b128 {frag(0, 0, i)}, v100 offset:{i * 2048}")
emit("1:")
for s in range(3): step(s)
emit("s_sub_u32 s24, s24, 3"); emit("s_cmp_lg_u32 s24, 0"); emit("s_cbranch_scc1 1b")
emit("s_waitcnt vmcnt(0) lgkmcnt(0)"); emit("s_nop 15"); emit("s_nop 15"); emit("s_memtime s[28:29]")
emit("v_accvgpr_read_b32 %[o0], a0"); emit("v_accvgpr_read_b32 %[o1], a127"); emit("s_waitcnt lgkmcnt(0)")
emit("s_sub_u32 s26, s28, s26"); emit("s_subb_u32 s27, s29, s27"); emit("v_mov_b32 %[t0], s26"); emit("v_mov_b32 %[t1], s27")
print("// generated by scripts/ubench/gen_duo3_loop.py --sched " + args.sched)
for line in out: print(f'"{line}\\n\\t"')
