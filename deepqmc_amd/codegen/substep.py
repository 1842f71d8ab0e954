"""Generator of PLAN-SPECIALISED Metropolis sub-step kernels (HIP source for gfx950) from a layer program.

The library's generic sub-step kernel (csrc/kernel_fused2.hip) INTERPRETS a list of unit descriptors per wave; three rounds of
tuning left it at ~104 us per sub-step of 4096 LiH / PauliNet walkers, latency bound: ~50 dependent phases per tile, each a
barrier, a descriptor fetch, an LDS round trip.  This module writes, for ONE program, a kernel in which all of that is
decided at generation time (reference semantics: sampling/electron_samplers.py:76-138, wf/nn_wave_function.py:127-173,
gnn/electron_gnn.py:160-276):

  * one WAVE owns a tile of 16 / N walkers (N = 4 electrons: 4 walkers) and keeps every activation in registers
    ("row-in-lanes", csrc/spec_device.h): layers chain MFMA -> MFMA through registers, sender gathers / spin means / sums over
    electrons are DPP permutations inside the 4-lane quad of a walker, no activation ever touches LDS, no barrier separates layers;
  * float32 products on the bf16 matrix pipe as in the generic kernel (operands split into three bf16 pieces, six
    v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block); the weights are pre-split and laid out on the host as a TAPE of 1 KB
    fragments in exactly the order the straight-line code consumes them (SpecTapeEntry, csrc/spec_device.h), streamed
    global -> registers -> LDS ring (3 stages of 16 KB) -> A operands by the four waves (= four tiles) of a workgroup together:
    one barrier per 16 KB, the ring's loads and stores spread between the MFMAs;
  * every K, width, offset, activation, edge pattern is a literal in the emitted source.

A program the generator does not cover raises Unsupported and keeps running on the generic kernel.  Covered: N = 4 electrons
with every edge row range a union of complete xor-offset blocks (2 up + 2 down), FEAT_EN / FEAT_EE / LINEAR (tanh, silu, none;
residuals; row-partitioned outputs) / CONV / SPIN_MEAN / ROW_SUM / ORBITALS + SLOGDET + FINAL tail -- the PauliNet / FermiNet
family on LiH-sized systems, which is BASELINE configs[0..1].
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

OP_FEAT_EN, OP_FEAT_EE, OP_LINEAR, OP_SPIN_MEAN, OP_CONV, OP_EDGE_SUM, OP_ROW_SUM = 1, 2, 3, 4, 5, 6, 7
OP_ORBITALS, OP_SLOGDET, OP_FINAL, OP_ATTENTION, OP_CONST = 8, 9, 10, 11, 12

STAGE_FRAGS = 16          # 1 KB fragments per ring stage
RING = 3
DEFAULTS = dict(
    pf=6,                 # tape fragments read ahead of their first use
    fold_tanh=1,          # 1: the factor 2 log2(e) of tanh(x) = 1 - 2 / (1 + exp2(x 2 log2 e)) is folded into the packed weights and bias
    barrier='asm',       # 'sync': __syncthreads() per ring stage; 'asm': s_waitcnt lgkmcnt(k) + s_barrier (the wave's own ring
                          #   stores are older than its last k LDS reads: no drain of the prefetched fragments)
    wpos=(1, 4, 7, 9),    # fragment positions inside a stage at which a quarter of the next stage is stored / the one after requested
    lazy_ring=1,          # 1: the first ring stage is stored and awaited at the first tape read, not in the prologue
    dpp_fold=0,           # 1: sender gathers / sums over electrons as v_fmac_f32_dpp / v_add_f32_dpp (one instruction) instead of v_mov_b32_dpp + op
    sgb=0,                # > 0: every scheduling region (between the sched_barriers of group_barrier) ends with a
                          #   sched_group_barrier pipeline {1 MFMA, sgb VALU} x its MFMAs
    defer_epi=0,          # 1: the epilogue (activation, residual, bf16 split) of a layer is interleaved with the MFMAs of the NEXT linear layer
                          #   when nothing in between reads its output (measured: 57.1 vs 57.0 us -- the wave is bound by instruction issue, not by order)
    share_means=1,        # 1: the per-walker (spin-mean) pieces of a wide layer are multiplied ONCE for the 16 walkers of a workgroup --
                          #   each wave takes a quarter of the output blocks with the walkers as the 16 MFMA columns -- instead of 4 x per tile
    producer=0,           # 1: a fifth wave per workgroup fills the weight ring (global -> registers -> LDS, one stage per barrier interval);
                          #   the four computing waves only read it
    waves=4,              # waves (= tiles of 4 walkers) per workgroup: 4 (one per SIMD) or 8 (two per SIMD)
    sched='',             # -amdgpu-sched-strategy of the translation unit ('' = the compiler's default)
    group_barrier=1,      # 1: __builtin_amdgcn_sched_barrier(0) in front of every batch of tape reads
)
N_OP_I = 28


class Unsupported(Exception):
    pass


def pad4(n: int) -> int:
    return (n + 3) // 4 * 4


def program_hash(n_up, n_down, n_nuc, n_det, bufs, ops, itable) -> int:
    """FNV-1a (64 bit) over the integers that define a program's structure; engine.hip computes the same value
    (Engine::program_hash).  Weights are not part of it: a specialised kernel serves every parameter set of its program."""
    h = 0xcbf29ce484222325

    def mix(v):
        nonlocal h
        for byte in struct.pack('<i', int(v)):
            h = ((h ^ byte) * 0x100000001b3) & 0xffffffffffffffff
    for v in (n_up, n_down, n_nuc, n_det, len(bufs), len(ops), len(itable)):
        mix(v)
    for (r, w) in bufs:
        mix(r); mix(w)
    for op in ops:
        mix(op.kind)
        ii = list(op.i) + [0] * (N_OP_I - len(op.i))
        for v in ii:
            mix(v)
        for fv in op.f:
            mix(struct.unpack('<i', struct.pack('<f', float(fv)))[0])
    for v in itable:
        mix(v)
    return h


@dataclass
class GBuf:
    """A program buffer (or one row of a per-walker buffer) as registers of the generated kernel."""
    name: str
    width: int                       # padded feature width
    layout: str                      # 'blocks' (MFMA output layout) or 'rep' (<= 8 features replicated in every lane group)
    rbs: Tuple[str, ...]             # row blocks: ('n',) for node / per-walker rows, edge blocks 'e1','e2','e3' (xor offset)
    has_f32: bool = False
    has_planes: bool = False
    lazy_mean: Optional[Tuple[str, int]] = None    # (source f32 array base name, 0 = up / 1 = down): planes made on demand

    @property
    def nb(self):
        return (self.width + 15) // 16

    @property
    def nch(self):
        return (self.nb + 1) // 2


class Gen:
    def __init__(self, name, n_up, n_down, n_nuc, n_det, bufs, ops, itable, **opts):
        self.name = name
        self.opt = dict(DEFAULTS)
        for k_, v_ in opts.items():
            if k_ not in self.opt:
                raise KeyError(k_)
            self.opt[k_] = v_
        self.n_up, self.n_down, self.n_nuc, self.K = n_up, n_down, n_nuc, n_det
        self.N = n_up + n_down
        self.bufs, self.ops, self.itable = bufs, ops, [int(v) for v in itable]
        if self.N != 4:
            raise Unsupported('row-in-lanes kernels are generated for 4 electrons (16 rows = 4 walkers per wave)')
        self.hash = program_hash(n_up, n_down, n_nuc, n_det, bufs, ops, itable)
        self.stream: List[Tuple[str, object]] = []      # ('c', text) | ('u', tape fragment index that must have been read)
        self.entries: List[dict] = []
        self.maps: List[int] = []
        self.n_frag = 0
        self.frag_kind: List[int] = []                  # per fragment: 0 weight plane (read into a register), 1 gather (read in place)
        self.g: Dict[Tuple[int, int], GBuf] = {}        # (program buffer, row or -1) -> registers
        self.pairs: Dict[int, List[Tuple[int, int]]] = {}
        self.uid = 0
        self.tail = None
        self.writers_done: Dict[int, int] = {}            # LINEAR ops emitted so far per destination buffer (row-partitioned outputs)
        self.conv_done: Dict[int, set] = {}               # blocks of a convolution's destination written so far
        self.xch_bytes = 0                              # LDS behind the ring: operand / result exchange of the shared mean products

    # ---- code stream ----
    def c(self, text):
        self.stream.append(('c', text))

    def use(self, frag):
        self.stream.append(('u', frag))

    def use_exact(self, frag):
        """ring cursor exactly past `frag` (no read-ahead): for code that reads fragments itself and therefore BEHIND the cursor --
        it may only touch the cursor's stage and the one before (the slot two stages back is being refilled)"""
        self.stream.append(('x', frag))

    def stamp(self, label):
        self.stream.append(('s', label))

    def fresh(self, p='v'):
        self.uid += 1
        return f'{p}{self.uid}'

    # ---- tape ----
    def tape_triple(self, w_off, ldw, col0, ncol, rows32, scale=1.0, explicit=False):
        """explicit: the generated code reads the fragments itself (a wave-dependent address), no automatic read ahead"""
        assert len(rows32) == 32
        self.entries.append(dict(kind=0, w_off=w_off, ldw=ldw, col0=col0, ncol=ncol, map=len(self.maps), scale=scale))
        self.maps.extend(rows32)
        f = self.n_frag
        self.n_frag += 3
        self.frag_kind += [2, 2, 2] if explicit else [0, 0, 0]
        return f

    def tape_gather(self, idx256, scale=1.0):
        assert len(idx256) == 256
        self.entries.append(dict(kind=1, w_off=0, ldw=0, col0=0, ncol=0, map=len(self.maps), scale=scale))
        self.maps.extend(idx256)
        f = self.n_frag
        self.n_frag += 1
        self.frag_kind.append(1)
        return f

    @staticmethod
    def lds_off(frag):
        s, o = divmod(frag, STAGE_FRAGS)
        return (s % RING) * STAGE_FRAGS * 1024 + o * 1024

    # ---- buffers ----
    def consumers(self, b):
        lin, other = 0, 0
        for op in self.ops:
            i = op.i
            if op.kind == OP_LINEAR:
                for p in range(i[0]):
                    if i[1 + 4 * p] == b:
                        lin += 1
                if i[25] == b:
                    other += 1
            elif op.kind in (OP_SPIN_MEAN, OP_ROW_SUM) and i[0] == b:
                other += 1
            elif op.kind == OP_CONV and (i[0] == b or i[1] == b):
                other += 1
            elif op.kind == OP_EDGE_SUM and i[0] == b:
                other += 1
            elif op.kind == OP_ORBITALS and i[0] == b:
                other += 1
            elif op.kind == OP_FINAL and i[0] == b:
                other += 1
        return lin, other

    def edge_blocks(self, b, r0, n):
        """xor-offset blocks covered by rows [r0, r0 + n) of edge buffer b (must be complete)."""
        prs = self.pairs[b][r0:r0 + n]
        by_d: Dict[int, set] = {}
        for (rc, sd) in prs:
            if sd < 0 or sd == rc:
                raise Unsupported('nuclear senders / self edges')
            by_d.setdefault(rc ^ sd, set()).add(rc)
        for d, s in by_d.items():
            if s != set(range(self.N)):
                raise Unsupported(f'edge rows [{r0},{r0 + n}) of buffer {b} do not form complete xor blocks (spin pattern)')
        if len(prs) != len(by_d) * self.N:
            raise Unsupported('duplicate edge rows')
        return tuple(f'e{d}' for d in sorted(by_d))

    def emit_split_blocks(self, gb: GBuf, rb, blocks):
        """planes of the chunks completed by `blocks` (f32 registers must hold them)"""
        for t in sorted({b // 2 for b in blocks}):
            b0, b1 = 2 * t, 2 * t + 1
            if b1 < gb.nb and not (b0 in blocks and b1 in blocks):
                continue
            src = [f'{gb.name}_{rb}[{b0}][{s}]' for s in range(4)] + ([f'{gb.name}_{rb}[{b1}][{s}]' for s in range(4)] if b1 < gb.nb else ['0.0f'] * 4)
            for j in range(4):
                if src[2 * j] == '0.0f':
                    self.c(f'{gb.name}p_{rb}[0][{t}].w[{j}] = 0u; {gb.name}p_{rb}[1][{t}].w[{j}] = 0u; {gb.name}p_{rb}[2][{t}].w[{j}] = 0u;')
                else:
                    self.c(f'split2({src[2 * j]}, {src[2 * j + 1]}, {gb.name}p_{rb}[0][{t}].w[{j}], {gb.name}p_{rb}[1][{t}].w[{j}], {gb.name}p_{rb}[2][{t}].w[{j}]);')

    def chunk_planes(self, gb: GBuf, rb, t):
        """names of the (hi, mid, lo) B operands of chunk t; lazily produced buffers emit their code here"""
        if gb.lazy_mean is not None:
            src, which = gb.lazy_mean
            e0, e1 = (0, self.n_up) if which == 0 else (self.n_up, self.N)
            inv = 1.0 / max(1, e1 - e0)
            v = self.fresh('mp')
            self.c(f'BfFrag {v}[3];')
            vals = []
            for bb in (2 * t, 2 * t + 1):
                for s in range(4):
                    if bb >= gb.nb:
                        vals.append('0.0f'); continue
                    x = f'{src}[{bb}][{s}]'
                    acc = f'quad_bcast<{e0}>({x})'        # (0 + x_e0) + x_e1 ... as the generic kernel's loop
                    for e in range(e0 + 1, e1):
                        acc = f'quad_add<{e}, {e}, {e}, {e}>({acc}, {x})' if self.opt['dpp_fold'] else f'({acc} + quad_bcast<{e}>({x}))'
                    vals.append(f'({acc} * {inv!r}f)' if e1 > e0 else '0.0f')
            for j in range(4):
                if vals[2 * j] == '0.0f':
                    self.c(f'{v}[0].w[{j}] = 0u; {v}[1].w[{j}] = 0u; {v}[2].w[{j}] = 0u;')
                else:
                    self.c(f'split2({vals[2 * j]}, {vals[2 * j + 1]}, {v}[0].w[{j}], {v}[1].w[{j}], {v}[2].w[{j}]);')
            return f'{v}[0]', f'{v}[1]', f'{v}[2]'
        assert gb.has_planes, gb.name
        return tuple(f'{gb.name}p_{rb}[{pl}][{t}]' for pl in range(3))

    # ---- ops ----
    def run(self):
        N = self.N
        # which ops form the network, which the tail
        kinds = [op.kind for op in self.ops]
        if kinds[-2:] != [OP_SLOGDET, OP_FINAL] or kinds[-3] != OP_ORBITALS:
            raise Unsupported('tail must be ORBITALS, SLOGDET, FINAL')
        for k, op in enumerate(self.ops[:-3]):
            self.stamp(f'op {k} kind {op.kind}')
            if op.kind == OP_FEAT_EN:
                self.op_feat_en(op)
            elif op.kind == OP_FEAT_EE:
                self.op_feat_ee(op)
            elif op.kind == OP_LINEAR:
                self.op_linear(k, op)
                self.stream.append(('ee', k))
            elif op.kind == OP_SPIN_MEAN:
                self.op_spin_mean(op)
            elif op.kind == OP_CONV:
                self.op_conv(op)
            elif op.kind == OP_ROW_SUM:
                self.op_row_sum(op)
            else:
                raise Unsupported(f'op kind {op.kind}')
        self.stamp('tail')
        self.op_tail(self.ops[-3], self.ops[-1])
        self.stamp('accept')

    def op_feat_en(self, op):
        b, lr, sp = op.i[0], op.i[1], op.i[2]
        width = self.bufs[b][1]
        nf = 4 * self.n_nuc + (1 if sp else 0)
        if nf > 8 or self.bufs[b][0] != self.N:
            raise Unsupported('electron-nucleus features wider than 8')
        gb = GBuf(f'b{b}', width, 'rep', ('n',))
        self.g[(b, -1)] = gb
        self.c(f'// op FEAT_EN -> buffer {b}: [|d|, d] per nucleus of the lane\'s electron (double, as pair_feature_lane)')
        self.c(f'float {gb.name}_n[8];')
        self.c('{ LaneInfo li1; li1.T = 1; li1.TP = 1; li1.N = 4;')
        for n in range(self.n_nuc):
            self.c(f'  {{ double dd[3] = {{(double)px - (double)a.R[{3 * n}], (double)py - (double)a.R[{3 * n + 1}], (double)pz - (double)a.R[{3 * n + 2}]}}, f4_[4];')
            self.c(f'    pair_feature_lane(dd, (double)a.eps, el, -1, 0, li1, {"true" if lr else "false"}, f4_);')
            self.c(f'    for (int q = 0; q < 4; ++q) {gb.name}_n[{4 * n} + q] = (float)f4_[q]; }}')
        c0 = 4 * self.n_nuc
        if sp:
            self.c(f'  {gb.name}_n[{c0}] = el < {self.n_up} ? 1.0f : -1.0f;')
            c0 += 1
        for q in range(c0, 8):
            self.c(f'  {gb.name}_n[{q}] = 0.0f;')
        self.c('}')
        gb.has_f32 = True

    def op_feat_ee(self, op):
        b, off, n_rows, lr = op.i[0], op.i[1], op.i[2], op.i[3]
        prs = [(self.itable[off + 2 * r], self.itable[off + 2 * r + 1]) for r in range(n_rows)]
        self.pairs[b] = prs
        rbs = self.edge_blocks(b, 0, n_rows)
        gb = GBuf(f'b{b}', self.bufs[b][1], 'rep', rbs)
        self.g[(b, -1)] = gb
        self.c(f'// op FEAT_EE -> buffer {b}: edge (receiver = the lane\'s electron, sender = electron ^ d), block per xor offset d')
        for rb in rbs:
            d = int(rb[1:])
            self.c(f'float {gb.name}_{rb}[8];')
            self.c(f'{{ LaneInfo li1; li1.T = 1; li1.TP = 1; li1.N = 4; double dd[3] = {{(double)px - (double)quad_xor<{d}>(px), (double)py - (double)quad_xor<{d}>(py), (double)pz - (double)quad_xor<{d}>(pz)}}, f4_[4];')
            self.c(f'  pair_feature_lane(dd, (double)a.eps, el, el ^ {d}, 0, li1, {"true" if lr else "false"}, f4_);')
            self.c(f'  for (int q = 0; q < 4; ++q) {{ {gb.name}_{rb}[q] = (float)f4_[q]; {gb.name}_{rb}[4 + q] = 0.0f; }} }}')
        gb.has_f32 = True

    def buf_of(self, b, row, bcast):
        """registers of program buffer b as a LINEAR piece / residual: whole node or edge buffer, or one per-walker row"""
        rows = self.bufs[b][0]
        if (b, -1) in self.g and not bcast:
            return self.g[(b, -1)]
        if (b, row) in self.g:
            return self.g[(b, row)]
        if (b, -1) in self.g and rows == 1:
            return self.g[(b, -1)]
        raise Unsupported(f'buffer {b} row {row} (bcast {bcast}) has no register form')

    def op_linear(self, k, op):
        i = op.i
        n_pieces, dst, dr0, dcol0, nrows, nout = i[0], i[17], i[18], i[19], i[20], i[21]
        w_off, bias_off, act, res, res_r0, normalize = i[22], i[23], i[24], i[25], i[26], i[27]
        if act not in (0, 1, 2):
            raise Unsupported(f'activation {act}')
        ldw = pad4(nout)
        nb = (ldw + 15) // 16
        if dcol0 % 16:
            raise Unsupported('destination column offset')
        # ---- inputs: row blocks and chunks ----
        pieces = []
        wrow = 0
        rbs = None
        node_rows = None        # (r0, n) when the op covers part of the node rows (row-partitioned output)
        for p in range(n_pieces):
            sb, sr0, Kp, bc = i[1 + 4 * p], i[2 + 4 * p], i[3 + 4 * p], i[4 + 4 * p]
            gb = self.buf_of(sb, sr0, bc)
            if gb.rbs != ('n',):
                prb = self.edge_blocks(sb, sr0, nrows)
                if rbs is not None and rbs != prb:
                    raise Unsupported('pieces over different edge blocks')
                rbs = prb
            else:
                if rbs is not None and rbs != ('n',):
                    raise Unsupported('node piece in an edge layer')
                rbs = ('n',)
                srows = self.bufs[sb][0]
                if not bc and srows == self.N and (sr0, nrows) != (0, self.N):
                    if node_rows is not None and node_rows != (sr0, nrows):
                        raise Unsupported('pieces over different node row ranges')
                    node_rows = (sr0, nrows)
            pieces.append((gb, Kp, wrow))
            wrow += pad4(Kp)
        # chunks: (planes per row block, 32 W rows)
        chunks = []
        rep_group = []       # consecutive 'rep' pieces share a chunk, one lane group each

        def flush_rep():
            nonlocal rep_group
            if not rep_group:
                return
            rows32 = [-1] * 32
            for gi, (gb, Kp, wr) in enumerate(rep_group):
                for s in range(8):
                    if s < Kp:
                        rows32[8 * gi + s] = wr + s
            chunks.append(('rep', list(rep_group), rows32))
            rep_group = []
        for (gb, Kp, wr) in pieces:
            if gb.layout == 'rep':
                if Kp > 8:
                    raise Unsupported('replicated piece wider than 8')
                rep_group.append((gb, Kp, wr))
                if len(rep_group) == 4:
                    flush_rep()
            else:
                flush_rep()
                for t in range(gb.nch):
                    rows32 = [-1] * 32
                    for gq in range(4):
                        for s in range(8):
                            f = 32 * t + 16 * (s >> 2) + 4 * gq + (s & 3)
                            if f < Kp:
                                rows32[8 * gq + s] = wr + f
                    if any(r >= 0 for r in rows32):
                        chunks.append(('blk', gb, t, rows32))
        flush_rep()
        # ---- destination ----
        drows = self.bufs[dst][0]
        dkey = (dst, -1)
        if rbs == ('n',):
            if drows not in (self.N, 1):
                raise Unsupported('linear layer into a multi-row per-walker buffer')
            d_rbs = ('n',)
        else:
            if dst not in self.pairs:
                self.pairs[dst] = [None] * drows
            src_b = i[1]
            for r in range(nrows):
                self.pairs[dst][dr0 + r] = self.pairs[src_b][i[2] + r]
            d_rbs = rbs
        n_lin, n_other = self.consumers(dst)
        dwidth = self.bufs[dst][1]
        need_f32 = n_other > 0 or node_rows is not None
        need_pl = n_lin > 0
        if dkey not in self.g:
            self.g[dkey] = GBuf(f'b{dst}', dwidth, 'blocks', ())
            self.g[dkey].has_f32, self.g[dkey].has_planes = need_f32, need_pl
            self.writers_done[dst] = 0
        gd = self.g[dkey]
        for rb in d_rbs:                      # registers of row blocks this op is the first to write
            if rb not in gd.rbs:
                gd.rbs = gd.rbs + (rb,)
                if gd.has_f32:
                    self.c(f'float {gd.name}_{rb}[{gd.nb}][4];')
                if gd.has_planes:
                    self.c(f'BfFrag {gd.name}p_{rb}[3][{gd.nch}];')
        first_writer = self.writers_done[dst] == 0
        self.writers_done[dst] += 1
        n_writers = sum(1 for o2 in self.ops if o2.kind == OP_LINEAR and o2.i[17] == dst)
        last_writer = self.writers_done[dst] == n_writers
        if dcol0 % 32 and need_pl:
            raise Unsupported('destination column offset inside a k-chunk')
        gres = None
        if res >= 0:
            gres = self.buf_of(res, res_r0, False)
            if not gres.has_f32:
                raise Unsupported('residual source without f32 registers')
        self.c(f'// op {k} LINEAR -> buffer {dst}: {len(chunks)} k-chunks x {nb} blocks x {len(rbs)} row block(s), act {act}' + (f', residual {res}' if res >= 0 else ''))
        tag = f'L{k}'
        wscale = 2.8853900817779268 if (act == 1 and self.opt['fold_tanh']) else 1.0      # 2 log2(e): tanh(x) = 1 - 2 / (1 + 2^(x 2 log2 e))
        # ---- bias fragments ----
        bias_frag = []
        if bias_off >= 0:
            for f0 in range(0, nb, 16):
                idx = [-1] * 256
                for bb in range(f0, min(nb, f0 + 16)):
                    for gq in range(4):
                        for s in range(4):
                            col = 16 * bb + 4 * gq + s
                            if col < nout:
                                idx[(bb - f0) * 16 + gq * 4 + s] = bias_off + col
                bias_frag.append(self.tape_gather(idx, wscale))
        dual = nb * len(rbs) == 1       # a lone accumulator would be one dependency chain: split it into small and large terms
        for rb in rbs:
            self.c(f'f32x4 {tag}b_{rb}[{nb}];' + (f' f32x4 {tag}s_{rb}[{nb}];' if dual else ''))
        if bias_frag:
            self.use(bias_frag[-1])
        for bb in range(nb):
            for rb in rbs:
                if bias_frag:
                    fo = self.lds_off(bias_frag[bb // 16]) + (bb % 16) * 64
                    self.c(f'{tag}b_{rb}[{bb}] = *reinterpret_cast<const f32x4*>(ring_g + {fo});')
                else:
                    self.c(f'{tag}b_{rb}[{bb}] = f32x4{{0, 0, 0, 0}};')
                if dual:
                    self.c(f'{tag}s_{rb}[{bb}] = f32x4{{0, 0, 0, 0}};')
        self.stream.append(('lb', k))
        # ---- per-walker pieces shared by the four tiles of the workgroup (option share_means) ----
        shared = [ch for ch in chunks if ch[0] == 'blk' and ch[1].lazy_mean is not None]
        do_share = bool(self.opt['share_means']) and self.opt['waves'] == 4 and rbs == ('n',) and nb % 4 == 0 and len(shared) >= 4
        if do_share:
            chunks = [ch for ch in chunks if not (ch[0] == 'blk' and ch[1].lazy_mean is not None)]
            nbw = nb // 4                                      # output blocks per wave
            xch = RING * STAGE_FRAGS * 1024                    # operand fragments [chunk][plane] x 1 KB, then results [block] x 1 KB
            res = xch + len(shared) * 3 * 1024
            self.xch_bytes = max(self.xch_bytes, len(shared) * 3 * 1024 + nb * 1024)
            self.c(f'// per-walker pieces once per workgroup: {len(shared)} k-chunks, wave w multiplies output blocks {nbw} w .. for all 16 walkers')
            self.c('{ const int xl_ = ((wave * 4 + (c >> 2)) + 16 * g) * 16;      // B-operand slot of this lane\'s walker (written by its el == 0 lanes)')
            for ci, ch in enumerate(shared):
                _, gb, t, rows32 = ch
                xh, xm, xl = self.chunk_planes(gb, 'n', t)
                for pl, v in enumerate((xh, xm, xl)):
                    self.c(f'  if (el == 0) st_frag(smem_raw + {xch + (ci * 3 + pl) * 1024} + xl_, {v});')
            self.c('}')
            self.c('__syncthreads();')
            for j in range(nbw):
                self.c(f'f32x4 {tag}m{j} = f32x4{{0, 0, 0, 0}};')
            def rd(ci):          # this wave's weight fragments of chunk ci + the chunk's shared operand
                # tape order (chunk, block-of-the-wave j, wave): the four waves' triples of one (chunk, j) are adjacent, so the
                # explicit reads lag the ring cursor by < one stage (the slot two stages back is being refilled)
                _, gb, t, rows32 = shared[ci]
                for j in range(nbw):
                    frs = []
                    for w in range(4):
                        bb = w * nbw + j
                        ncol = max(0, min(16, nout - 16 * bb))
                        frs.append(self.tape_triple(w_off, ldw, 16 * bb, ncol, rows32, wscale, explicit=True))
                    self.use_exact(frs[-1] + 2)      # (12 fragments back at most: < one stage)
                    for pl in range(3):
                        o = [self.lds_off(frs[w] + pl) for w in range(4)]
                        self.c(f'const BfFrag {tag}w{ci}_{j}_{pl} = ld_frag(ring_lane + (wave_s == 0 ? {o[0]} : wave_s == 1 ? {o[1]} : wave_s == 2 ? {o[2]} : {o[3]}));')
                for pl in range(3):
                    self.c(f'const BfFrag {tag}x{ci}_{pl} = ld_frag(smem_raw + {xch + (ci * 3 + pl) * 1024} + lane * 16);')
            rd(0)
            for ci in range(len(shared)):
                if ci + 1 < len(shared):
                    rd(ci + 1)
                for (wp, xp) in [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]:
                    for j in range(nbw):
                        self.c(f'{tag}m{j} = mfma_bf16({tag}w{ci}_{j}_{wp}, {tag}x{ci}_{xp}, {tag}m{j});')
            for j in range(nbw):
                self.c(f'*reinterpret_cast<f32x4*>(smem_raw + {res} + (wave * {nbw} + {j}) * 1024 + lane * 16) = {tag}m{j};')
            self.c('__syncthreads();')
        # ---- products: chunk outer, blocks inner in pairs; the six products of a block (small terms first) alternate with
        # those of its partner, so consecutive MFMAs never share an accumulator ----
        prods = [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]      # (weight plane, activation plane)
        for ci, ch in enumerate(chunks):
            ops_b = {}
            if len(chunks) * nb >= 64:
                self.stamp(f'op {k} chunk {ci}')
            if ch[0] == 'rep':
                _, group, rows32 = ch
                for rb in rbs:
                    v = self.fresh('rp')
                    self.c(f'BfFrag {v}[3];')
                    for j in range(4):
                        def sel(s):
                            e = '0.0f'
                            for gi in reversed(range(len(group))):
                                gb, Kp, wr = group[gi]
                                src = f'{gb.name}_{rb if gb.rbs != ("n",) else "n"}[{s}]' if s < Kp else '0.0f'
                                e = f'(g == {gi} ? {src} : {e})'
                            return e
                        self.c(f'split2({sel(2 * j)}, {sel(2 * j + 1)}, {v}[0].w[{j}], {v}[1].w[{j}], {v}[2].w[{j}]);')
                    ops_b[rb] = (f'{v}[0]', f'{v}[1]', f'{v}[2]')
            else:
                _, gb, t, rows32 = ch
                for rb in rbs:
                    ops_b[rb] = self.chunk_planes(gb, rb if gb.rbs != ('n',) else 'n', t)
            for b0 in range(0, nb, 2):
                grp = [bb for bb in (b0, b0 + 1) if bb < nb]
                fr = {}
                for bb in grp:
                    ncol = max(0, min(16, nout - 16 * bb))
                    fr[bb] = self.tape_triple(w_off, ldw, 16 * bb, ncol, rows32, wscale)
                self.use(fr[grp[-1]] + 2)
                for (wp, xp) in prods:
                    for bb in grp:
                        for rb in rbs:
                            acc = f'{tag}s_{rb}[{bb}]' if (dual and wp + xp == 2) else f'{tag}b_{rb}[{bb}]'
                            self.c(f'{acc} = mfma_bf16(t{fr[bb] + wp}, {ops_b[rb][xp]}, {acc});')
        self.stream.append(('le', k))
        if do_share:
            self.c('{ const int xr_ = ((wave * 4 + (c >> 2)) + 16 * g) * 16;')
            for bb in range(nb):
                self.c(f'  {tag}b_n[{bb}] += *reinterpret_cast<const f32x4*>(smem_raw + {res} + {bb * 1024} + xr_);')
            self.c('}')
        if len(chunks) * nb >= 64:
            self.stamp(f'op {k} epilogue')
        # ---- epilogue ----
        self.stream.append(('eb', k))
        scale = '0.70710678118654752440f' if normalize else None
        for rb in rbs:
            drb = rb
            for bb in range(nb):
                db = bb + dcol0 // 16
                for s in range(4):
                    e = f'({tag}s_{rb}[{bb}][{s}] + {tag}b_{rb}[{bb}][{s}])' if dual else f'({tag}b_{rb}[{bb}][{s}])'
                    if act == 1:
                        e = f'tanh_scaled{e}' if wscale != 1.0 else f'tanh_value{e}'
                    elif act == 2:
                        e = f'silu_value{e}'
                    if gres is not None:
                        rrb = rb if gres.rbs != ('n',) else 'n'
                        e = f'({gres.name}_{rrb}[{db}][{s}] + {e})'
                        if scale:
                            e = f'({e} * {scale})'
                    if node_rows is not None and not first_writer:
                        r0, n = node_rows
                        e = f'((el >= {r0} && el < {r0 + n}) ? {e} : {gd.name}_{drb}[{db}][{s}])'
                    if gd.has_f32:
                        self.c(f'{gd.name}_{drb}[{db}][{s}] = {e};')
                    else:
                        if s == 0:
                            self.c(f'float {tag}o_{rb}_{bb}[4];')
                        self.c(f'{tag}o_{rb}_{bb}[{s}] = {e};')
            if gd.has_planes:
                if node_rows is not None:
                    if not gd.has_f32:
                        raise Unsupported('row-partitioned output without f32 registers')
                    if last_writer:
                        self.emit_split_blocks(gd, drb, set(range(gd.nb)))
                elif gd.has_f32:
                    self.emit_split_blocks(gd, drb, set(range(dcol0 // 16, dcol0 // 16 + nb)))
                else:
                    for t in range((nb + 1) // 2):
                        b0, b1 = 2 * t, 2 * t + 1
                        src = [f'{tag}o_{rb}_{b0}[{s}]' for s in range(4)] + ([f'{tag}o_{rb}_{b1}[{s}]' for s in range(4)] if b1 < nb else ['0.0f'] * 4)
                        tt = t + dcol0 // 32
                        for j in range(4):
                            if src[2 * j] == '0.0f':
                                self.c(f'{gd.name}p_{drb}[0][{tt}].w[{j}] = 0u; {gd.name}p_{drb}[1][{tt}].w[{j}] = 0u; {gd.name}p_{drb}[2][{tt}].w[{j}] = 0u;')
                            else:
                                self.c(f'split2({src[2 * j]}, {src[2 * j + 1]}, {gd.name}p_{drb}[0][{tt}].w[{j}], {gd.name}p_{drb}[1][{tt}].w[{j}], {gd.name}p_{drb}[2][{tt}].w[{j}]);')

    # (the marker that closes the epilogue is appended by run())

    def op_spin_mean(self, op):
        src, dst = op.i[0], op.i[1]
        gs = self.g[(src, -1)]
        if gs.rbs != ('n',):
            raise Unsupported('spin mean of an edge buffer')
        self.c(f'// op SPIN_MEAN buffer {src} -> {dst}: produced where the consuming layer needs it (DPP inside the walker\'s quad)')
        for which in (0, 1):
            e0, e1 = (0, self.n_up) if which == 0 else (self.n_up, self.N)
            if gs.layout == 'rep':
                gm = GBuf(f'b{dst}r{which}', gs.width, 'rep', ('n',))
                inv = 1.0 / max(1, e1 - e0)
                self.c(f'float {gm.name}_n[8];')
                for s in range(8):
                    x = f'{gs.name}_n[{s}]'
                    if e1 > e0:
                        acc = f'quad_bcast<{e0}>({x})'
                        for e in range(e0 + 1, e1):
                            acc = f'({acc} + quad_bcast<{e}>({x}))'
                        self.c(f'{gm.name}_n[{s}] = {acc} * {inv!r}f;')
                    else:
                        self.c(f'{gm.name}_n[{s}] = 0.0f;')
                gm.has_f32 = True
            else:
                if not gs.has_f32:
                    raise Unsupported('spin mean of a buffer without f32 registers')
                gm = GBuf(f'b{dst}r{which}', gs.width, 'blocks', ('n',), lazy_mean=(f'{gs.name}_n', which))
            self.g[(dst, which)] = gm

    def op_conv(self, op):
        we, hx, dst, col0, tab, S, W = op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], op.i[6]
        gw, gh = self.g[(we, -1)], self.g[(hx, -1)]
        if gh.rbs != ('n',) or not (gw.has_f32 and gh.has_f32) or W % 16 or col0 % 16 or gw.layout != 'blocks' or gh.layout != 'blocks':
            raise Unsupported('convolution operands')
        ds = None
        for el in range(self.N):
            dd = []
            for s in range(S):
                row, snd = self.itable[tab + 2 * (el * S + s)], self.itable[tab + 2 * (el * S + s) + 1]
                if row < 0:
                    raise Unsupported('ragged sender lists')
                if self.pairs[we][row] != (el, snd) or snd < 0:
                    raise Unsupported('convolution table does not match the edge rows')
                dd.append(el ^ snd)
            if ds is None:
                ds = sorted(dd)
            elif sorted(dd) != ds:
                raise Unsupported('sender pattern differs between receivers')
        n_lin, n_other = self.consumers(dst)
        if (dst, -1) not in self.g:
            gd = GBuf(f'b{dst}', self.bufs[dst][1], 'blocks', ('n',))
            self.g[(dst, -1)] = gd
            self.c(f'float {gd.name}_n[{gd.nb}][4];')
            gd.has_f32 = True
            if n_lin:
                self.c(f'BfFrag {gd.name}p_n[3][{gd.nch}];')
                gd.has_planes = True
        gd = self.g[(dst, -1)]
        self.c(f'// op CONV: buffer {dst} cols {col0}.. = sum over senders of w(edge) * h(sender), senders = electron ^ {ds}')
        blocks = set()
        for bb in range(W // 16):
            db = bb + col0 // 16
            blocks.add(db)
            for s in range(4):
                e = None
                for d in ds:
                    if self.opt['dpp_fold']:
                        e = f'quad_fmac<{0 ^ d}, {1 ^ d}, {2 ^ d}, {3 ^ d}>({e if e is not None else "0.0f"}, {gw.name}_e{d}[{bb}][{s}], {gh.name}_n[{bb}][{s}])'
                        continue
                    term = f'{gw.name}_e{d}[{bb}][{s}] * quad_xor<{d}>({gh.name}_n[{bb}][{s}])'
                    e = f'(0.0f + {term})' if e is None else f'({e} + {term})'
                self.c(f'{gd.name}_n[{db}][{s}] = {e};')
        if gd.has_planes:
            # chunks whose two blocks are both written by now
            done = self.conv_done.setdefault(dst, set())
            done |= blocks
            ready = {b for b in blocks if (b ^ 1) in done or (b ^ 1) >= gd.nb}
            self.emit_split_blocks(gd, 'n', ready | {b ^ 1 for b in ready if (b ^ 1) < gd.nb})

    def op_row_sum(self, op):
        src, dst = op.i[0], op.i[1]
        gs = self.g[(src, -1)]
        if gs.rbs != ('n',) or not gs.has_f32 or gs.layout != 'blocks':
            raise Unsupported('row sum operand')
        n_lin, n_other = self.consumers(dst)
        gd = GBuf(f'b{dst}', gs.width, 'blocks', ('n',))
        self.g[(dst, -1)] = gd
        self.c(f'// op ROW_SUM buffer {src} -> {dst} (every lane of the quad gets the sum over the walker\'s electrons)')
        self.c(f'float {gd.name}_n[{gd.nb}][4];')
        gd.has_f32 = True
        for bb in range(gd.nb):
            for s in range(4):
                x = f'{gs.name}_n[{bb}][{s}]'
                acc = f'(0.0f + quad_bcast<0>({x}))'
                for e in range(1, self.N):
                    acc = f'quad_add<{e}, {e}, {e}, {e}>({acc}, {x})' if self.opt['dpp_fold'] else f'({acc} + quad_bcast<{e}>({x}))'
                self.c(f'{gd.name}_n[{bb}][{s}] = {acc};')
        if n_lin:
            self.c(f'BfFrag {gd.name}p_n[3][{gd.nch}];')
            gd.has_planes = True
            self.emit_split_blocks(gd, 'n', set(range(gd.nb)))

    def op_tail(self, orb, fin):
        N, K, n_nuc = self.N, self.K, self.n_nuc
        bfb = orb.i[0]
        n_env = orb.i[6] if orb.i[6] > 0 else 1
        o_pu, o_pd, o_zu, o_zd = orb.i[2], orb.i[3], orb.i[4], orb.i[5]
        gbf = self.g[(bfb, -1)]
        if not gbf.has_f32 or gbf.layout != 'blocks':
            raise Unsupported('backflow buffer')
        nbk = (K * N + 15) // 16
        if nbk > 4:
            raise Unsupported('more than 16 determinants per walker tile (one determinant per lane)')
        M = N * n_nuc * n_env                 # envelope parameters per (lane, block): [mu][nucleus][envelope]
        nfr = (M + 3) // 4
        self.c('// ---- tail: Slater matrices = envelope * backflow (the arithmetic of k_orbitals, value lane) ----')
        self.c(f'float sm[{nbk}][4];       // row `el` of the matrix of determinant 4 b + g')
        frags = {}
        for bb in range(nbk):
            for which in (0, 1):        # pi, zeta
                for q in range(nfr):
                    idx = [-1] * 256
                    for lane in range(64):
                        c_, gq = lane & 15, lane >> 4
                        el = c_ & 3
                        kd = 4 * bb + gq
                        if kd >= K:
                            continue
                        base = (o_pu if el < self.n_up else o_pd) if which == 0 else (o_zu if el < self.n_up else o_zd)
                        for j in range(4):
                            m = 4 * q + j
                            if m < M:
                                mu, rest = divmod(m, n_nuc * n_env)
                                idx[lane * 4 + j] = base + (kd * N + mu) * n_nuc * n_env + rest
                    frags[(bb, which, q)] = self.tape_gather(idx)
        self.use(max(frags.values()))
        self.c(f'{{ float rho[{n_nuc}];')
        for n in range(n_nuc):
            self.c(f'  {{ const float dx = px - a.R[{3 * n}], dy = py - a.R[{3 * n + 1}], dz = pz - a.R[{3 * n + 2}]; float d2 = a.eps; d2 += dx * dx; d2 += dy * dy; d2 += dz * dz; rho[{n}] = sqrtf(d2); }}')
        for bb in range(nbk):
            self.c(f'  {{ float pi_[{4 * nfr}], ze_[{4 * nfr}];')
            for q in range(nfr):
                self.c(f'    {{ const f32x4 t_ = *reinterpret_cast<const f32x4*>(ring_lane + {self.lds_off(frags[(bb, 0, q)])}); for (int j = 0; j < 4; ++j) pi_[{4 * q} + j] = t_[j]; }}')
                self.c(f'    {{ const f32x4 t_ = *reinterpret_cast<const f32x4*>(ring_lane + {self.lds_off(frags[(bb, 1, q)])}); for (int j = 0; j < 4; ++j) ze_[{4 * q} + j] = t_[j]; }}')
            for mu in range(N):
                self.c(f'    {{ float acc_ = 0.f;')
                for n in range(n_nuc):
                    for ev in range(n_env):
                        m = (mu * n_nuc + n) * n_env + ev
                        self.c(f'      acc_ += pi_[{m}] * expf(-fabsf(ze_[{m}]) * rho[{n}]);')
                self.c(f'      sm[{bb}][{mu}] = acc_ * {gbf.name}_n[{bb}][{mu}]; }}')
            self.c('  }')
        self.c('}')
        # transpose inside the quad: lane e ends with the whole matrix of block b = e
        self.c('// 4 x 4 transposition of (block, lane) inside the quad: afterwards lane e holds rows 0..3 of the matrix of determinant 4 e + g')
        if nbk < 4:
            self.c(f'float smz[4][4]; for (int b_ = 0; b_ < 4; ++b_) for (int s_ = 0; s_ < 4; ++s_) smz[b_][s_] = b_ < {nbk} ? sm[b_ < {nbk} ? b_ : 0][s_] : 0.0f;')
            arr = 'smz'
        else:
            arr = 'sm'
        for kbit in (0, 1):
            for b0 in range(4):
                if (b0 >> kbit) & 1:
                    continue
                b1 = b0 | (1 << kbit)
                for s in range(4):
                    self.c(f'{{ const bool hi_ = (el >> {kbit}) & 1; const float snd_ = hi_ ? {arr}[{b0}][{s}] : {arr}[{b1}][{s}]; const float rcv_ = quad_xor<{1 << kbit}>(snd_); if (hi_) {arr}[{b0}][{s}] = rcv_; else {arr}[{b1}][{s}] = rcv_; }}')
        self.c(f'const int kd = 4 * el + g;')
        self.c(f'double la = 0.0; int sn = 0;')
        self.c(f'{{ float m_[16]; for (int i_ = 0; i_ < 4; ++i_) for (int j_ = 0; j_ < 4; ++j_) m_[4 * i_ + j_] = {arr}[i_][j_];')
        self.c(f'  if (kd < {K}) fused2_det<float, 4>(m_, la, sn); }}')
        jas, cc_off, cusp_kind, al_off = fin.i[0], fin.i[1], fin.i[2], fin.i[3]
        same_scale, anti_scale = float(fin.f[0]), float(fin.f[1])
        self.c('// exp-normalised CI sum over the 16 lanes of the walker (wf/nn_wave_function.py:152-160); lane (e, g) <-> determinant 4 e + g')
        self.c(f'double shift = kd < {K} ? la : -INFINITY;')
        self.c('shift = fmax(shift, __shfl_xor(shift, 16, 64)); shift = fmax(shift, __shfl_xor(shift, 32, 64)); shift = fmax(shift, __shfl_xor(shift, 1, 64)); shift = fmax(shift, __shfl_xor(shift, 2, 64));')
        self.c('if (isinf(shift)) shift = 0.0;')
        ccx = f'(double)a.w[{cc_off} + (kd < {K} ? kd : 0)]' if cc_off >= 0 else '1.0'
        self.c(f'double psi = kd < {K} ? {ccx} * sn * exp(la - shift) : 0.0;')
        self.c('psi += __shfl_xor(psi, 16, 64); psi += __shfl_xor(psi, 32, 64); psi += __shfl_xor(psi, 1, 64); psi += __shfl_xor(psi, 2, 64);')
        self.c('double logpsi = log(fabs(psi)) + shift;')
        self.c('const int sign_p = (psi > 0) - (psi < 0);')
        if cusp_kind:
            n_pairs = N * (N - 1) // 2
            pr = [(i_, j_) for i_ in range(N) for j_ in range(i_ + 1, N)]
            self.c('// electron cusps (wf/cusp.py:5-26,68-78): pair p on lane 4 e + g = p of the walker, summed like the CI terms')
            self.c('{ const int lw = 4 * el + g;')
            isel = ' : '.join(f'lw == {p} ? {pr[p][0]}' for p in range(n_pairs)) + ' : 0'
            jsel = ' : '.join(f'lw == {p} ? {pr[p][1]}' for p in range(n_pairs)) + ' : 1'
            self.c(f'  const int ci_ = {isel}, cj_ = {jsel};')
            self.c('  const int qb_ = (threadIdx.x & 63) & 12;')
            self.c('  const float xi = __shfl(px, qb_ | ci_, 64), yi = __shfl(py, qb_ | ci_, 64), zi = __shfl(pz, qb_ | ci_, 64);')
            self.c('  const float xj = __shfl(px, qb_ | cj_, 64), yj = __shfl(py, qb_ | cj_, 64), zj = __shfl(pz, qb_ | cj_, 64);')
            self.c('  double cusp = 0.0;')
            self.c(f'  if (lw < {n_pairs}) {{')
            self.c('    double d2 = (double)a.eps; { const double d_ = (double)xi - (double)xj; d2 += d_ * d_; } { const double d_ = (double)yi - (double)yj; d2 += d_ * d_; } { const double d_ = (double)zi - (double)zj; d2 += d_ * d_; }')
            self.c('    const double rho_ = sqrt(d2);')
            self.c(f'    const bool same_ = (ci_ < {self.n_up}) == (cj_ < {self.n_up});')
            self.c(f'    const double sc_ = same_ ? {same_scale!r} : {anti_scale!r}, alp_ = (double)a.w[{al_off} + (same_ ? 0 : 1)];')
            if cusp_kind == 1:
                self.c('    cusp = -sc_ / (alp_ * (1 + alp_ * rho_));')
            else:
                self.c('    cusp = -sc_ * alp_ * alp_ / (alp_ + rho_);')
            self.c('  }')
            self.c('  cusp += __shfl_xor(cusp, 16, 64); cusp += __shfl_xor(cusp, 32, 64); cusp += __shfl_xor(cusp, 1, 64); cusp += __shfl_xor(cusp, 2, 64);')
            self.c('  logpsi += cusp;')
            self.c('}')
        else:
            self.c('logpsi += 0.0;')
        if jas >= 0:
            gj = self.g[(jas, -1)]
            if not gj.has_f32:
                raise Unsupported('Jastrow buffer')
            self.c(f'logpsi += (double)__shfl({gj.name}_n[0][0], (threadIdx.x & 63) & 15, 64);')
        self.tail = True

    @staticmethod
    def op_rw(op):
        i = op.i
        if op.kind in (OP_FEAT_EN, OP_FEAT_EE, OP_CONST):
            return set(), {i[0]}
        if op.kind == OP_LINEAR:
            rd = {i[1 + 4 * p] for p in range(i[0])}
            if i[25] >= 0:
                rd.add(i[25])
            return rd, {i[17]}
        if op.kind in (OP_SPIN_MEAN, OP_ROW_SUM):
            return {i[0]}, {i[1]}
        if op.kind == OP_CONV:
            return {i[0], i[1]}, {i[2]}
        if op.kind == OP_EDGE_SUM:
            return {i[0]}, {i[2]}
        if op.kind == OP_ORBITALS:
            return {i[0]}, {i[1]}
        if op.kind == OP_FINAL:
            return ({i[0]} if i[0] >= 0 else set()), set()
        return set(), set()

    def can_defer(self, k, k2):
        """may the epilogue of LINEAR op k run inside the product loop of LINEAR op k2 > k?"""
        rd_k, wr_k = self.op_rw(self.ops[k])
        dst = self.ops[k].i[17]
        for j in range(k + 1, k2 + 1):
            rd, wr = self.op_rw(self.ops[j])
            if dst in rd:
                return False
            if j < k2 and (wr & (rd_k | {dst})):
                return False
            # spin means are produced where they are consumed: a consumer of a mean of dst reads dst
            if self.ops[j].kind == OP_LINEAR:
                for p in range(self.ops[j].i[0]):
                    sb = self.ops[j].i[1 + 4 * p]
                    for o2 in self.ops[:j]:
                        if o2.kind == OP_SPIN_MEAN and o2.i[1] == sb and o2.i[0] == dst:
                            return False
        return True

    def defer_epilogues(self):
        st = self.stream
        out = []
        i = 0
        while i < len(st):
            it = st[i]
            if it[0] == 'eb':
                k = it[1]
                j = next(n for n in range(i, len(st)) if st[n] == ('ee', k))
                E = st[i + 1:j]
                n = next((q for q in range(j + 1, len(st)) if st[q][0] == 'lb'), None)
                if self.opt['defer_epi'] and n is not None and self.can_defer(k, st[n][1]):
                    k2 = st[n][1]
                    m = next(q for q in range(n, len(st)) if st[q] == ('le', k2))
                    body = st[n + 1:m]
                    mf = [q for q, b in enumerate(body) if b[0] == 'c' and 'mfma_bf16(' in b[1]]
                    if mf and E:
                        out.extend(st[j + 1:n])
                        per = -(-len(E) // len(mf))
                        nb_, e0 = [], 0
                        for q, b in enumerate(body):
                            nb_.append(b)
                            if q in mf and e0 < len(E):
                                nb_.extend(E[e0:e0 + per])
                                e0 += per
                        nb_.extend(E[e0:])
                        out.extend(nb_)
                        i = m
                        continue
                out.extend(E)
                i = j + 1
                continue
            out.append(it)
            i += 1
        self.stream = [x for x in out if x[0] not in ('lb', 'le', 'eb', 'ee')]

    def add_group_pipelines(self, lines):
        res, region = [], []

        def flush():
            m = sum(1 for l in region if 'mfma_bf16(' in l)
            v = sum(5 * l.count('tanh_') + 11 * l.count('split2(') + 2 * l.count('quad_') for l in region)
            res.extend(region)
            if m >= 2 and v >= 4:
                q = min(int(self.opt['sgb']), -(-v // m))
                for _ in range(m):
                    res.append(f'  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, {q}, 0);')
            region.clear()
        for l in lines:
            if '__builtin_amdgcn_sched_barrier(0);' in l:
                flush()
                res.append(l)
            else:
                region.append(l)
        res.extend(region)
        return res

    # ---- final assembly ----
    def source(self):
        self.run()
        self.defer_epilogues()
        n_stage = (self.n_frag + STAGE_FRAGS - 1) // STAGE_FRAGS
        tape_bytes = n_stage * STAGE_FRAGS * 1024
        lds_bytes = RING * STAGE_FRAGS * 1024 + self.xch_bytes
        out: List[str] = []
        A = out.append
        kname = f'k_substep_{self.name}'
        A('// hipcc-flags:' + (f' -mllvm -amdgpu-sched-strategy={self.opt["sched"]}' if self.opt['sched'] else ''))
        A(f'// GENERATED by deepqmc_amd/codegen/substep.py for program hash 0x{self.hash:016x} -- do not edit; regenerate with')
        A('//   python -m deepqmc_amd.codegen')
        A('// One Metropolis sub-step (propose -> psi -> accept) of a tile of 4 walkers per wave, activations in registers, weights from')
        A('// the tape through an LDS ring shared by the 4 waves of the workgroup (csrc/spec_device.h; reference')
        A('// sampling/electron_samplers.py:76-138, wf/nn_wave_function.py:127-173).')
        A('#include "../spec_device.h"')
        A('namespace dqmc {')
        A('namespace {')
        A('using namespace dqmc::spec;')
        nw = self.opt['waves']
        prod = bool(self.opt['producer'])
        nthr = 64 * nw
        npc = (STAGE_FRAGS * 64) // nthr      # 16-byte pieces per thread and ring stage
        nthr_launch = nthr + (64 if prod else 0)
        wpos = list(self.opt['wpos'])[:npc] if len(self.opt['wpos']) >= npc else list(self.opt['wpos'])
        if len(wpos) != npc:
            raise ValueError('wpos must name one position per piece')
        A(f'template <bool PROF> __global__ void __launch_bounds__({nthr_launch}) {kname}(const SpecArgs a) {{')
        A('  HIP_DYNAMIC_SHARED(char, smem_raw)')
        A('  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;')
        A('  const int c = lane & 15, g = lane >> 4, wl = c >> 2, el = c & 3;')
        A('  const int wave_s = __builtin_amdgcn_readfirstlane(wave);')
        A('  (void)wave_s;')
        A('  (void)wl;')
        A(f'  const long bw = (long)blockIdx.x * {4 * nw} + wave * 4 + (c >> 2);      // this lane\'s walker')
        A('  const bool live = bw < a.B;')
        A('  const long bc = live ? bw : (long)a.B - 1;')
        A('  const char* ring_lane = smem_raw + lane * 16;')
        A('  const char* ring_g = smem_raw + g * 16;')
        A('  V16* ring_t = reinterpret_cast<V16*>(smem_raw) + tid;')
        A(f'  const TapeRsrc tape = tape_rsrc(a.tape, {tape_bytes});')
        A('  const int tid16 = tid * 16;')
        A('@@PRODUCER@@')
        A('  const bool stamp_ = PROF && a.prof != nullptr && blockIdx.x == 0 && lane == 0;')
        A('  if (stamp_) a.prof[wave * 256] = clock64();')
        if not prod:
            A('  // ring prologue: stage 0 -> LDS, stage 1 -> registers')
            A('  V16 ' + ', '.join(f'rq{q}' for q in range(npc)) + ';')
            for q in range(npc):
                A(f'  rq{q} = tape_load<{q * nthr * 16}>(tape, tid16);')
        # sampler state + proposal while the first stage is in flight
        A('  // step size of this sub-step from the previous one\'s acceptance (the arithmetic of k_tau_update)')
        A('  float tau;')
        A('  if (a.mc.s == 0) {')
        A('    tau = reinterpret_cast<const float*>(a.mc.tau_in)[0];')
        A('  } else {')
        A('    tau = reinterpret_cast<const float*>(a.mc.tau_ring)[(a.mc.s + 1) & 1];')
        A('    if (a.mc.target > 0) {')
        A('      const float acceptance = (float)a.mc.counters[(a.mc.s + 2) % 3] / (float)a.B;')
        A('      const float m_ = acceptance > 0.05f ? acceptance : 0.05f;')
        A('      tau = tau / ((float)a.mc.target / m_);')
        A('    }')
        A('  }')
        A('  if (blockIdx.x == 0 && tid == 0) {')
        A('    reinterpret_cast<float*>(a.mc.tau_ring)[a.mc.s & 1] = tau;')
        A('    a.mc.counters[(a.mc.s + 1) % 3] = 0;')
        A('  }')
        A('  // lane (walker, electron, g < 3) proposes coordinate g: r\' = r + tau xi (electron_samplers.py:102-104)')
        A('  float rp = 0.0f;')
        A('  if (g < 3) {')
        A('    const long o_ = bc * 12 + el * 3 + g;')
        A('    rp = reinterpret_cast<const float*>(a.mc.r)[o_] + tau * reinterpret_cast<const float*>(a.mc.noise)[o_];')
        A('  }')
        A('  const float lp_old = reinterpret_cast<const float*>(a.mc.logpsi)[bc];')
        A('  const float u_b = reinterpret_cast<const float*>(a.mc.unif)[bc];')
        A('  const int age_b = a.mc.age[bc];')
        A('  const float px = __shfl(rp, c, 64), py = __shfl(rp, c + 16, 64), pz = __shfl(rp, c + 32, 64);')
        events = []      # the workgroup barriers of the computing waves, in order: what the producer wave has to match

        def ring_start():
            if prod:
                events.append(('start',))
                A('  __syncthreads();')
                return
            for q in range(npc):
                A(f'  ring_t[{q * nthr}] = rq{q};')
            if n_stage > 1:
                for q in range(npc):
                    A(f'  rq{q} = tape_load<{(STAGE_FRAGS * 64 + q * nthr) * 16}>(tape, tid16);')
            A('  __syncthreads();')
        started = False
        if not self.opt['lazy_ring']:
            ring_start(); started = True
        # body with the tape reads hoisted PF fragments ahead
        emitted = 0

        # LDS reads a wave issues between its last ring store of a stage and the barrier that ends the stage
        def reads_after_last_store(s):
            return sum(1 for f2 in range(s * STAGE_FRAGS + wpos[-1], min(self.n_frag, (s + 1) * STAGE_FRAGS)) if self.frag_kind[f2] == 0)

        def emit_read(f):
            nonlocal started
            if not started:
                ring_start(); started = True
            s, o = divmod(f, STAGE_FRAGS)
            if o == 0 and s >= 1 and prod:
                events.append(('ring', s))
                A('  ring_barrier0();')
            elif o == 0 and s >= 1:
                kk = min(6, reads_after_last_store(s - 1))
                if self.opt['barrier'] == 'asm' and kk >= 1:
                    A(f'  ring_barrier<{kk}>();')
                else:
                    A('  __syncthreads();')
            if o in wpos and not prod:
                q = wpos.index(o)
                if s + 1 < n_stage:
                    A(f'  ring_t[{((s + 1) % RING) * STAGE_FRAGS * 64 + q * nthr}] = rq{q};')
                if s + 2 < n_stage:
                    A(f'  rq{q} = tape_load<{((s + 2) * STAGE_FRAGS * 64 + q * nthr) * 16}>(tape, tid16);')
            if self.frag_kind[f] == 0:
                A(f'  const BfFrag t{f} = ld_frag(ring_lane + {self.lds_off(f)});')
        n_stamp = 0
        self.stamp_labels = []
        for kind, v in self.stream:
            if kind == 's':
                n_stamp += 1
                self.stamp_labels.append(v)
                A(f'  if (stamp_) a.prof[wave * 256 + {n_stamp}] = clock64();      // before {v}')
            elif kind in ('u', 'x'):
                target = min(self.n_frag, int(v) + 1 + (self.opt['pf'] if kind == 'u' else 0))
                if self.opt['group_barrier'] and emitted < target:
                    A('  __builtin_amdgcn_sched_barrier(0);')
                while emitted < target:
                    emit_read(emitted)
                    emitted += 1
            else:
                if '__syncthreads();' in v:
                    events.append(('x',))
                A('  ' + v)
        while emitted < self.n_frag:
            emit_read(emitted)
            emitted += 1
        # accept / reject (k_accept), sampler state, acceptance count
        A('  // accept = 2 (log|psi\'| - log|psi|) > log u  [| age >= max_age]; identical in the 16 lanes of a walker')
        A('  const float lp_prop = (float)logpsi;')
        A('  const float log_prob = 2 * (lp_prop - lp_old);')
        A('  bool acc = log_prob > logf(u_b);')
        A('  if (a.mc.max_age >= 0) acc = acc || (age_b >= a.mc.max_age);')
        A('  if (live) {')
        A('    if (acc && g < 3) reinterpret_cast<float*>(a.mc.r)[bw * 12 + el * 3 + g] = rp;')
        A('    if (el == 0 && g == 0) {')
        A('      if (acc) {')
        A('        reinterpret_cast<float*>(a.mc.logpsi)[bw] = lp_prop;')
        A('        a.mc.sign[bw] = sign_p;')
        A('        a.mc.age[bw] = 0;')
        A('      } else {')
        A('        a.mc.age[bw] = age_b + 1;')
        A('      }')
        A('      if (a.mc.accept_out) a.mc.accept_out[bw] = acc ? 1 : 0;')
        A('    }')
        A('  }')
        A('  int n_acc = (live && acc && el == 0 && g == 0) ? 1 : 0;')
        A('  n_acc += __shfl_xor(n_acc, 4, 64); n_acc += __shfl_xor(n_acc, 8, 64);')
        A('  if (lane == 0 && n_acc) atomicAdd(a.mc.counters + a.mc.s % 3, n_acc);')
        A(f'  if (stamp_) a.prof[wave * 256 + {n_stamp + 1}] = clock64();')
        A('}')
        pl = []
        if prod:
            P = pl.append
            P(f'  if (wave == {nw}) {{      // the producer wave: stage s + 1 goes into the ring while the computing waves read stage s')
            P('    char* ring_w = smem_raw + lane * 16;')
            P('    const int l16 = lane * 16;')

            def copy_stage(st_):
                if st_ >= n_stage:
                    return
                P(f'    {{ V16 v_[{STAGE_FRAGS}];')
                for q in range(STAGE_FRAGS):
                    P(f'      v_[{q}] = tape_load<{(st_ * STAGE_FRAGS + q) * 1024}>(tape, l16);')
                for q in range(STAGE_FRAGS):
                    P(f'      *reinterpret_cast<V16*>(ring_w + {(st_ % RING) * STAGE_FRAGS * 1024 + q * 1024}) = v_[{q}];')
                P('    }')
            for ev in events:
                if ev[0] == 'start':
                    copy_stage(0)
                    copy_stage(1)
                    P('    __syncthreads();')
                elif ev[0] == 'ring':
                    P('    __syncthreads();')
                    copy_stage(ev[1] + 1)
                else:
                    P('    __syncthreads();')
            P('    return;')
            P('  }')
        k_ = out.index('@@PRODUCER@@')
        out[k_:k_ + 1] = pl
        if self.opt['sgb']:
            out[:] = self.add_group_pipelines(out)
        A(f'void launch_{self.name}(hipStream_t st, const SpecArgs& a, int n_blocks) {{')
        A(f'  static const bool lds_ok_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&{kname}<true>), hipFuncAttributeMaxDynamicSharedMemorySize, {lds_bytes}) == hipSuccess &&')
        A(f'                             hipFuncSetAttribute(reinterpret_cast<const void*>(&{kname}<false>), hipFuncAttributeMaxDynamicSharedMemorySize, {lds_bytes}) == hipSuccess;')
        A('  (void)lds_ok_;')
        A(f'  if (a.prof) hipLaunchKernelGGL(HIP_KERNEL_NAME({kname}<true>), dim3((unsigned)n_blocks), dim3({nthr_launch}), {lds_bytes}, st, a);')
        A(f'  else hipLaunchKernelGGL(HIP_KERNEL_NAME({kname}<false>), dim3((unsigned)n_blocks), dim3({nthr_launch}), {lds_bytes}, st, a);')
        A('}')
        A(f'const SpecTapeEntry entries_{self.name}[] = {{')
        for e in self.entries:
            A(f'  {{{e["kind"]}, {e["w_off"]}, {e["ldw"]}, {e["col0"]}, {e["ncol"]}, {e["map"]}, {e["scale"]!r}f}},')
        A('};')
        A(f'const int32_t maps_{self.name}[] = {{')
        for k in range(0, len(self.maps), 32):
            A('  ' + ', '.join(str(v) for v in self.maps[k:k + 32]) + ',')
        A('};')
        A(f'const char* const stamps_{self.name} = "' + '|'.join(self.stamp_labels) + '";')
        A('}  // namespace')
        A(f'const SpecKernel* spec_kernel_{self.name}() {{')
        A(f'  static const SpecKernel k = {{0x{self.hash:016x}ull, "{self.name}", {4 * nw}, {lds_bytes}, {tape_bytes}, {len(self.entries)}, entries_{self.name}, maps_{self.name}, launch_{self.name}, stamps_{self.name}}};')
        A('  return &k;')
        A('}')
        A('}  // namespace dqmc')
        return '\n'.join(out) + '\n'


def generate(name, program, **opts) -> str:
    """HIP source of the specialised sub-step kernel of `program` (deepqmc_amd.program.Program)."""
    g = Gen(name, program.n_up, program.n_down, program.n_nuc, program.spec.n_determinants, program.bufs, program.ops, program.itable, **opts)
    return g.source()
