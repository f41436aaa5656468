"""Host side of the one-launch REDUCE front end (pynqs_amd/csrc/kernels_reduce_onepass.hip, include/pynqs_amd.h:
pynqs_reduce_onepass / pynqs_reduce_contract): the buffers, their capacities and the two launches.

What the reference does between the walkers and the ansatz in `_reduce_psi` (vmc/energy/eloc.py:205-324) and `Func`
(vmc/energy/flip.py:29-63) -- get_comb_tensor, get_hij_torch, the |H| >= eps mask, torch.multinomial on the sub-eps part,
WavefunctionLUT.lookup, torch.unique(dim=0), onv_to_tensor, the scatter back and the contraction -- is here
    front.run(x, plan, eps, seed, lut)          one kernel: records + distinct x' (+-1 rows ready for the ansatz)
    psi_u = ansatz(front.uniq_pm1[:n_unique])   the caller's module
    eloc, psi_x = front.contract(psi_u, ...)    one kernel
with nothing read back in between.  Buffers have fixed capacities (a HIP graph can hold the whole step); what a call would have
NEEDED comes back in `counters`, so an overflow is noticed whenever the caller next looks (energy.local_energy looks once per call,
when it needs the number of distinct rows to slice the ansatz' input; a captured step looks after the replay).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _native as N

OVERFLOW_DOUBLES, OVERFLOW_TABLE, OVERFLOW_UNIQUE = 1, 2, 4
ROW_CACHE = True                  # semi-stochastic kernel: cache the row in global memory for the draws (see ReduceFrontEnd)
ROW_CACHE_MAX_BYTES = 8 << 30
ROW_F32 = __import__("os").environ.get("PYNQS_ROW_F32", "1") != "0"  # semi-stochastic calls on short rows: the two-kernel form (see ReduceFrontEnd)
ROW_F32_MAX_BYTES = int(__import__("os").environ.get("PYNQS_ROW_F32_MAX_BYTES", str(16 << 30)))  # long rows: the float32 copy is 4 bytes per column and walker
TILE_SCRATCH_MIN_ROW = int(__import__("os").environ.get("PYNQS_TILE_SCRATCH_MIN_ROW", "65536"))  # columns per row from which the tile sums leave the LDS


def _pow2_at_least(v: int) -> int:
    c = 64
    while c < v:
        c <<= 1
    return c


def geometry(n: int, sorb: int, nele: int, noa: int, nob: int, eps_sample: int, dedup_slots: int = 0) -> Tuple[int, int, int, bool]:
    """(segments, fixed slots per segment, bytes of a de-duplication table of `dedup_slots` slots, fused form available)."""
    out = (C.c_int64 * 4)(0, 0, dedup_slots, 0)
    N.check(N.lib().pynqs_reduce_onepass_geometry(n, sorb, nele, noa, nob, int(eps_sample), out), "pynqs_reduce_onepass_geometry")
    return int(out[0]), int(out[1]), int(out[2]), bool(out[3])


def supported(n: int, sorb: int, nele: int, noa: int, nob: int, eps_sample: int) -> bool:
    if sorb % 2 or not 0 <= eps_sample <= 65535:
        return False
    try:
        return geometry(n, sorb, nele, noa, nob, eps_sample)[3]
    except RuntimeError:
        return False


def wants_row_cache(n: int, ncomb: int, eps_sample: int, nchunks: int, esz: int) -> bool:
    """the semi-stochastic kernel's row cache: worth it when the draws are dense in the row (>= one draw per 64 columns) and the
    [n, ncomb] scratch is affordable (<= ROW_CACHE_MAX_BYTES)"""
    return bool(eps_sample > 0 and nchunks == 1 and ROW_CACHE and eps_sample * 64 >= ncomb and n * ncomb * esz <= ROW_CACHE_MAX_BYTES)


def list_capacity(n: int, sorb: int, nele: int, noa: int, nob: int, eps_sample: int, h_dtype: torch.dtype = torch.float64,
                  without_table: bool = False) -> int:
    """Largest cap_doubles with which the kernel keeps a segment's records in an LDS list (its LIST form or the flushing form;
    pynqs_reduce_onepass_list_capacity), or -1.  without_table: for a front end built with dedup=False."""
    ncomb = int(N.lib().pynqs_num_sd(sorb, noa, nob)) + 1
    esz = 8 if h_dtype == torch.float64 else 4
    nseg = geometry(n, sorb, nele, noa, nob, eps_sample)[0]
    cache = wants_row_cache(n, ncomb, int(eps_sample), nseg // max(n, 1) if n else 1, esz)
    out = C.c_int64(-1)
    N.check(N.lib().pynqs_reduce_onepass_list_capacity(n, sorb, nele, noa, nob, N.PYNQS_F64 if esz == 8 else N.PYNQS_F32, int(eps_sample),
                                                       int(cache), int(without_table), C.byref(out)), "pynqs_reduce_onepass_list_capacity")
    return int(out.value)


class ReduceFrontEnd:
    """Buffers of the fused REDUCE front end for `n` walkers of one system.  Capacities: `cap_doubles` compacted slots per
    (walker, chunk) segment for the kept doubles (column 0, the singles and the unpaired doubles have fixed slots), `cap_unique`
    rows for the distinct x'."""

    def __init__(self, n: int, sorb: int, nele: int, noa: int, nob: int, eps_sample: int, h_dtype: torch.dtype, device,
                 cap_doubles: int, cap_unique: int, pm1_dtype: torch.dtype = torch.float64, keep_onv: bool = True, want_pm1: bool = True,
                 dedup: bool = True) -> None:
        if h_dtype not in (torch.float64, torch.float32) or pm1_dtype not in (torch.float64, torch.float32):
            raise TypeError("float32 / float64 only")
        self.n, self.sorb, self.nele, self.noa, self.nob = int(n), sorb, nele, noa, nob
        self.eps_sample = int(eps_sample)
        self.h_dtype, self.pm1_dtype, self.device = h_dtype, pm1_dtype, torch.device(device)
        self.L = (sorb - 1) // 64 + 1
        self.cap_doubles, self.cap_unique = max(int(cap_doubles), 0), max(int(cap_unique), 1)
        self.dedup_slots = _pow2_at_least(2 * self.cap_unique)
        self.nseg, self.fixed, table_bytes, ok = geometry(self.n, sorb, nele, noa, nob, self.eps_sample, self.dedup_slots)
        if not ok:
            raise RuntimeError("row too long for the fused REDUCE front end (LDS): use reduce_compact / reduce_compact_sampled")
        self.nchunks = self.nseg // max(self.n, 1) if self.n else 1
        self.stride = self.fixed + self.cap_doubles
        dev, L = self.device, self.L
        slots = max(self.nseg * self.stride, 1)
        self.rec_col = torch.empty(slots, dtype=torch.int32, device=dev)
        self.rec_w = torch.zeros(slots, dtype=h_dtype, device=dev)
        self.rec_onv = torch.zeros((slots, 8 * L), dtype=torch.uint8, device=dev) if keep_onv else None
        self.rec_link = torch.empty(slots, dtype=torch.int32, device=dev)
        self.seg_count = torch.zeros(max(self.nseg, 1), dtype=torch.int32, device=dev)
        ns = max(self.n * self.eps_sample, 1)
        self.srec_col = torch.empty(ns, dtype=torch.int32, device=dev)
        self.srec_w = torch.zeros(ns, dtype=h_dtype, device=dev)
        self.srec_onv = torch.zeros((ns, 8 * L), dtype=torch.uint8, device=dev) if keep_onv else None
        self.srec_link = torch.empty(ns, dtype=torch.int32, device=dev)
        self.row_sum = torch.zeros(max(self.n, 1), dtype=torch.float64, device=dev)
        # dedup = False: no de-duplication table -- every record gets its own row of the distinct list (uniq_onv then holds duplicates and
        # `counters[0]` counts records); for systems whose x' are nearly all distinct, where the table would be gigabytes of random probes
        self.dedup = bool(dedup)
        self.table = torch.empty(table_bytes // 4, dtype=torch.int32, device=dev) if self.dedup else None
        self.slot_i32 = table_bytes // 4 // self.dedup_slots
        self.row_off = 2 if L == 1 else 1
        self.uniq_onv = torch.zeros((self.cap_unique, 8 * L), dtype=torch.uint8, device=dev)
        # rows beyond the distinct count keep whatever an earlier call (or the first walker, below) left there: always a valid +-1 row,
        # so that a captured step may run the ansatz on all cap_unique rows
        # (want_pm1 = False: the caller evaluates the amplitudes from the packed determinants, e.g. pynqs_rbm_forward: 0.1 ms and 0.47 GB of
        # writes less per 1.5 M distinct x')
        self.uniq_pm1 = torch.ones((self.cap_unique, sorb), dtype=pm1_dtype, device=dev) if want_pm1 else None
        self.uniq_parent = torch.zeros(self.cap_unique, dtype=torch.int32, device=dev)  # the walker each distinct row descends from
        self.counters = torch.zeros(4, dtype=torch.int32, device=dev)
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)  # added to run()'s seed: bump it between the replays of a captured step
        # row cache of the semi-stochastic kernel: the draws read the row back instead of enumerating the drawn tiles again
        ncomb = int(N.lib().pynqs_num_sd(sorb, noa, nob)) + 1
        esz = 8 if h_dtype == torch.float64 else 4
        self.row_cache = None
        # the two-kernel semi-stochastic form (round 4; rows of up to 32768 columns): the enumerating kernel leaves the row's sub-eps elements as
        # float32 here, the draw kernel reads them once into registers; neither the float64 row cache nor the tile scratch is needed then
        # (2: the flushing form -- rows of any length, kept columns beyond the LDS list -- whose draws read the drawn tiles back from the same
        # float32 copy instead of enumerating them again; on long rows it still keeps its tile sums in the tile scratch)
        self.row_f32, self.row_f32_form = None, 0
        if self.eps_sample > 0 and ROW_F32:
            self.row_f32_form = int(N.lib().pynqs_reduce_onepass_wants_row_f32(
                self.n, sorb, nele, noa, nob, N.PYNQS_F64 if esz == 8 else N.PYNQS_F32, self.eps_sample, self.cap_doubles,
                int(ncomb > TILE_SCRATCH_MIN_ROW), int(not self.dedup)))
            if self.row_f32_form in (1, 2):
                nel = int(N.lib().pynqs_reduce_onepass_row_f32_elements(self.n, sorb, nele, noa, nob))
                if self.row_f32_form == 1 or 4 * nel <= ROW_F32_MAX_BYTES:
                    self.row_f32 = torch.empty(max(nel, 1), dtype=torch.float32, device=dev)
        if self.row_f32 is None and wants_row_cache(self.n, ncomb, self.eps_sample, self.nchunks, esz):
            self.row_cache = torch.empty(max(self.n * ncomb, 1), dtype=h_dtype, device=dev)
        # long rows with draws and no row cache: the tile sums / draw counts of the kernel in global memory instead of the LDS
        self.tile_scratch = None
        if self.eps_sample > 0 and self.row_cache is None and (self.row_f32 is None or self.row_f32_form == 2) and ncomb > TILE_SCRATCH_MIN_ROW:
            nb = int(N.lib().pynqs_reduce_onepass_tile_scratch_bytes(self.n, sorb, nele, noa, nob, self.eps_sample))
            if nb > 0:
                self.tile_scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
        self._lut = None
        self._io = self._make_io()

    def _make_io(self, rec_w: Optional[Tensor] = None, srec_w: Optional[Tensor] = None, lut=None) -> N.ReduceIO:
        io = N.ReduceIO()
        io.cap_doubles, io.cap_unique, io.dedup_slots = self.cap_doubles, self.cap_unique, self.dedup_slots
        io.rec_col, io.rec_w = self.rec_col.data_ptr(), (self.rec_w if rec_w is None else rec_w).data_ptr()
        io.rec_onv = self.rec_onv.data_ptr() if self.rec_onv is not None else None
        io.rec_link, io.seg_count = self.rec_link.data_ptr(), self.seg_count.data_ptr()
        io.srec_col, io.srec_w = self.srec_col.data_ptr(), (self.srec_w if srec_w is None else srec_w).data_ptr()
        io.srec_onv = self.srec_onv.data_ptr() if self.srec_onv is not None else None
        io.srec_link, io.row_sum = self.srec_link.data_ptr(), self.row_sum.data_ptr()
        io.dedup_table, io.uniq_onv = (self.table.data_ptr() if self.table is not None else None), self.uniq_onv.data_ptr()
        io.uniq_pm1 = self.uniq_pm1.data_ptr() if self.uniq_pm1 is not None else None
        io.pm1_dtype = N.PYNQS_F64 if self.pm1_dtype == torch.float64 else N.PYNQS_F32
        io.lut_is_hash = 1
        io.lut_table = lut.table.data_ptr() if lut is not None else None
        io.lut_nkeys = lut.nkeys if lut is not None else 0
        io.counters = self.counters.data_ptr()
        io.seed_dev = self.seed_dev.data_ptr()
        io.row_cache = self.row_cache.data_ptr() if self.row_cache is not None else None
        io.uniq_parent = self.uniq_parent.data_ptr()
        io.tile_scratch = self.tile_scratch.data_ptr() if self.tile_scratch is not None else None
        io.tile_scratch_bytes = self.tile_scratch.numel() if self.tile_scratch is not None else 0
        io.row_f32 = self.row_f32.data_ptr() if self.row_f32 is not None else None
        return io

    def record_stream(self, stream) -> None:
        """The buffers are (also) used on `stream`: when this front end was allocated under another stream's context (total_energy's
        look-ahead), the caching allocator must wait for that stream's pending work before it hands the memory out again."""
        for t in vars(self).values():
            if isinstance(t, Tensor) and t.is_cuda:
                t.record_stream(stream)

    # ---- launches -------------------------------------------------------------------------------------------------------
    def run(self, x: Tensor, plan: Tensor, eps: float, seed: int = 0, lut=None) -> None:
        """Enqueue the front end for the walkers `x` (uint8 [n, 8 len]).  `lut`: the hash table of a WavefunctionLUT (its
        `.hashtable`): determinants found there take their amplitude from the table and do not enter the distinct list."""
        if x.size(0) != self.n or not x.is_cuda or x.dtype != torch.uint8 or not x.is_contiguous():
            raise RuntimeError("x must be a contiguous uint8 CUDA tensor with the front end's number of walkers")
        code = N.PYNQS_F64 if self.h_dtype == torch.float64 else N.PYNQS_F32
        if lut is not self._lut:
            self._lut, self._io = lut, self._make_io(lut=lut)
        st = torch.cuda.current_stream(self.device).cuda_stream
        N.check(N.lib().pynqs_reduce_onepass(x.data_ptr(), self.n, self.sorb, self.nele, self.noa, self.nob, plan.data_ptr(), code,
                                             float(eps), self.eps_sample, int(seed) & (2**64 - 1), C.byref(self._io), st),
                "pynqs_reduce_onepass")

    def contract(self, values_unique: Tensor, values_table: Optional[Tensor] = None, divide: bool = True,
                 rec_w: Optional[Tensor] = None, srec_w: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        """(sum_records w A(x') [/ A(x)], A(x)) per walker; A = `values_unique` on the distinct rows, `values_table` on the
        wave-function table's positions.  rec_w / srec_w replace the records' weights (same slot layout)."""
        cplx = values_unique.is_complex()
        vt = torch.complex128 if cplx else torch.float64
        vu = values_unique.to(vt).contiguous()
        # (the kernel only reads the rows the records name: values_unique must cover the distinct rows of the last run, not cap_unique)
        tb = values_table.to(vt).contiguous() if values_table is not None else None
        if self._lut is not None and tb is None:
            raise RuntimeError("the records refer to a wave-function table: pass its values")
        io = self._io if rec_w is None and srec_w is None else self._make_io(rec_w, srec_w, self._lut)
        out = torch.empty(self.n, dtype=vt, device=self.device)
        ax = torch.empty(self.n, dtype=vt, device=self.device)
        code = N.PYNQS_F64 if (rec_w if rec_w is not None else self.rec_w).dtype == torch.float64 else N.PYNQS_F32
        st = torch.cuda.current_stream(self.device).cuda_stream
        N.check(N.lib().pynqs_reduce_contract(self.n, self.sorb, self.nele, self.noa, self.nob, code, self.eps_sample, C.byref(io),
                                              vu.data_ptr(), tb.data_ptr() if tb is not None else None, int(cplx), int(divide),
                                              out.data_ptr(), ax.data_ptr(), st), "pynqs_reduce_contract")
        return out, ax

    # ---- what came back -------------------------------------------------------------------------------------------------
    def counters_host(self) -> Tuple[int, int, int]:
        """(distinct rows needed, overflow bits, largest kept-doubles count of a segment): ONE device-to-host read (synchronises)."""
        c = self.counters.tolist()
        return int(c[0]), int(c[1]), int(c[2])

    def overflowed(self, counters: Optional[Tuple[int, int, int]] = None) -> bool:
        nu, flags, mx = counters if counters is not None else self.counters_host()
        return bool(flags) or nu > self.cap_unique or mx > self.cap_doubles

    def table_rows(self) -> Tensor:
        """int32 view [dedup_slots]: the distinct-list row of every de-duplication slot (-1: empty)."""
        return self.table.view(self.dedup_slots, self.slot_i32)[:, self.row_off]

    def rows_of(self, link: Tensor) -> Tensor:
        """Distinct-list rows (int64) of record links >= 0: direct links (>= 2^30) carry the row, the others the de-duplication slot."""
        link = link.long()
        direct = link >= (1 << 30)
        rows = torch.where(direct, link - (1 << 30), torch.zeros_like(link))
        if not bool(direct.all()):
            rows = torch.where(direct, rows, self.table_rows().long()[torch.where(direct, torch.zeros_like(link), link)])
        return rows

    def count_records(self) -> int:
        """number of valid kept + drawn records of the last call (synchronises)"""
        col = self.rec_col.view(self.nseg, self.stride)
        slot = torch.arange(self.stride, device=self.device).unsqueeze(0)
        kept = self.seg_count[: self.nseg].clamp(max=self.cap_doubles).unsqueeze(1)
        m = int(torch.where(slot < self.fixed, col >= 0, slot < self.fixed + kept).sum())
        if self.eps_sample > 0:
            m += int((self.srec_col[: self.n * self.eps_sample] >= 0).sum())
        return m

    def records(self):
        """Flat view of the valid records, for the host-side algebra of the projected / multi-psi forms and for tests (synchronises):
        (walker int64[m], col int32[m], w[m], link int32[m], onv uint8[m, 8 len] or None, is_drawn bool[m]); kept records first
        (segment by segment, slot order), then the drawn ones."""
        dev = self.device
        col = self.rec_col.view(self.nseg, self.stride)
        slot = torch.arange(self.stride, device=dev).unsqueeze(0)
        kept = self.seg_count[: self.nseg].clamp(max=self.cap_doubles).unsqueeze(1)
        valid = torch.where(slot < self.fixed, col >= 0, slot < self.fixed + kept).reshape(-1)
        idx = torch.nonzero(valid).squeeze(1)
        walker = (idx // self.stride) // self.nchunks
        parts = [(walker, self.rec_col[idx], self.rec_w[idx], self.rec_link[idx], self.rec_onv[idx] if self.rec_onv is not None else None)]
        if self.eps_sample > 0:
            sidx = torch.nonzero(self.srec_col[: self.n * self.eps_sample] >= 0).squeeze(1)
            parts.append((sidx // self.eps_sample, self.srec_col[sidx], self.srec_w[sidx], self.srec_link[sidx],
                          self.srec_onv[sidx] if self.srec_onv is not None else None))
        cat = lambda k: torch.cat([p[k] for p in parts])  # noqa: E731
        drawn = torch.cat([torch.zeros(p[0].numel(), dtype=torch.bool, device=dev) if i == 0 else torch.ones(p[0].numel(), dtype=torch.bool, device=dev)
                           for i, p in enumerate(parts)])
        return cat(0), cat(1), cat(2), cat(3), (cat(4) if self.rec_onv is not None else None), drawn


class ReduceStep:
    """REDUCE local energies of a fixed batch shape with nothing read back on the way:
        front end (one kernel) -> `amplitude` on ALL cap_unique rows of the distinct list -> contraction (one kernel).
    The shapes are static (rows beyond the distinct count hold valid +-1 rows whose amplitudes nobody reads), so the whole step can
    be replayed from a HIP graph (graph=True: captured on first use; the draw seed lives in device memory and is bumped inside the
    graph).  `check()` is the one read-back: call it whenever convenient (e.g. with the energy statistics of the step); it raises
    OverflowError when a buffer was too small -- the results of that step must then be discarded and the step rebuilt larger."""

    def __init__(self, front: ReduceFrontEnd, plan: Tensor, eps: float, amplitude, lut=None, lut_values: Optional[Tensor] = None,
                 seed: int = 0, graph: bool = False) -> None:
        self.front, self.plan, self.eps, self.amplitude = front, plan, float(eps), amplitude
        self.lut, self.lut_values, self.seed = lut, lut_values, int(seed)
        self.x = torch.zeros((front.n, 8 * front.L), dtype=torch.uint8, device=front.device)
        self.eloc = self.psi_x = None
        self._graph = None
        self._want_graph = graph

    def _body(self) -> None:
        fe = self.front
        fe.run(self.x, self.plan, self.eps, self.seed, self.lut)
        if fe.eps_sample:
            fe.seed_dev.add_(1)
        with torch.no_grad():
            psi_u = self.amplitude(fe.uniq_pm1)
        self.eloc, self.psi_x = fe.contract(psi_u, self.lut_values)

    def __call__(self, x: Tensor) -> Tuple[Tensor, Tensor]:
        self.x.copy_(x)
        if not self._want_graph:
            self._body()
        elif self._graph is None:
            # warm up on a side stream (allocator pools, lazily loaded kernels), then capture
            s = torch.cuda.Stream(self.front.device)
            s.wait_stream(torch.cuda.current_stream(self.front.device))
            with torch.cuda.stream(s):
                for _ in range(2):
                    self._body()
            torch.cuda.current_stream(self.front.device).wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._body()
            self._graph = g
            g.replay()
        else:
            self._graph.replay()
        return self.eloc, self.psi_x

    def check(self) -> Tuple[int, int, int]:
        cnt = self.front.counters_host()
        if self.front.overflowed(cnt):
            raise OverflowError(f"REDUCE front end too small: needed {cnt[0]} distinct rows (capacity {self.front.cap_unique}), "
                                f"{cnt[2]} kept columns in a segment (capacity {self.front.cap_doubles}), flags {cnt[1]}")
        return cnt
