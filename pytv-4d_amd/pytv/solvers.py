"""Device-resident drivers of the reference's user-level loops.

The reference ships its solvers as README snippets that round-trip numpy <-> GPU four times per
iteration (README.md:118-124 sub-gradient descent, :141-157 Chambolle-Pock; ADMM is only mentioned,
:26,135).  Here the state lives on the GPU(s) for the whole run and every iteration is a handful of
fused HIP kernels (include/pytv4d.h):

    ChambollePock       tv_cp_fused + tv_cp_fixup (one sweep: q read and written once)  (5+2Nd) words/voxel + fix-up
                        or the pair tv_cp_dual (D + sigma-step + projection, TV partial) (1+2Nd)
                        + tv_cp_primal (fidelity dual + D^T + primal step, loss partial)  (Nd+5)
    ChambollePockOperator  the same with a user data-fidelity operator A / A^T on device tensors (tv_cp_dual + tv_cpop_*)
    ADMM                tv_admm_tu / tv_admm_zu, tv_DT_axpy, tv_normal_op2 (I + rho D^T D from x alone, two dot products) +
                        tv_cg_update (single-reduction CG; textbook tv_cg_step1/2 kept)
    SubgradientDescent  tv_subgrad_step_fused (TV, G and the descent step in ONE pass over x) or tv_subgrad + tv_subgrad_step

With a ``Slab`` (one process per GPU) each rank holds a contiguous z-slab; one boundary plane per
neighbour is exchanged per operator apply (two for the radius-2 kernels) and, for Chambolle-Pock,
hidden behind the interior planes' kernel.  Scalars (TV, fidelity, CG dots) stay on the device as
fp64 and are all-reduced there; nothing synchronises with the host inside the loop.
"""

import torch

from . import _native as _nv
from .slab import HaloPlan, Slab

__all__ = ["ChambollePock", "ChambollePockOperator", "ADMM", "SubgradientDescent", "cp_step_size", "auto_pitch"]


def cp_step_size(nz_global, m, reg_z_over_reg, reg_time, time_weight_max=1.0):
    """tau = 1 / (1 + L^2) with L^2 = 4 (2 + reg_z [z active] + reg_time * time_weight_max [t active]) >= |D|^2.
    time_weight_max: the largest per-pixel weight on the time regularisation (``factor_reg_static`` where
    ``mask_static`` is set, the maximum of a weight map; ``Geometry.time_weight_max``) -- the time channels are scaled
    by its square root, so it belongs in the bound.  Reduces to the reference's 1/(8+1) in 2-D (README.md:143)."""
    z = nz_global > 1 and reg_z_over_reg > 0
    t = m > 1 and reg_time > 0
    return 1.0 / (1.0 + 4.0 * (2.0 + (reg_z_over_reg if z else 0.0) + (reg_time * time_weight_max if t else 0.0)))


def normal_spectral_bound(scheme, nz_global, m, reg_z_over_reg, reg_time, time_weight_max=1.0):
    """L >= lambda_max(D^T D): 4 per one-sided difference axis (1 per halved central one) times the square of the axis weight;
    hybrid is the mean of the two one-sided operators.  1 / (1 + L) is cp_step_size for the one-sided schemes."""
    z = nz_global > 1 and reg_z_over_reg > 0
    t = m > 1 and reg_time > 0
    s = 2.0 + (reg_z_over_reg if z else 0.0) + (reg_time * time_weight_max if t else 0.0)
    return (1.0 if scheme == "central" else 4.0) * s


def chebyshev_coefficients(lmax, n):
    """(alpha_k, beta_k), k = 0 .. n-1, of e_{k+1} = e_k + alpha_k (b - A e_k) + beta_k (e_k - e_{k-1}), e_0 = e_{-1} = 0, for a
    symmetric A with spectrum in [1, lmax] (Saad, Iterative Methods for Sparse Linear Systems, alg. 12.1 as a three-term
    recurrence).  The scalars depend on the interval only: no dot product, no all-reduce."""
    theta, delta = 0.5 * (lmax + 1.0), 0.5 * (lmax - 1.0)
    if delta <= 1e-14 * theta:
        return [(1.0 / theta, 0.0)] * n
    sigma = theta / delta
    out, rho_prev = [(1.0 / theta, 0.0)], 1.0 / sigma
    for _ in range(1, n):
        rho_k = 1.0 / (2.0 * sigma - rho_prev)
        out.append((2.0 * rho_k / delta, rho_k * rho_prev))
        rho_prev = rho_k
    return out[:n]


def auto_pitch(ny, nx, dtype, frame_pad_bytes=0):
    """(row_pitch, frame_pitch) in elements for a solver's private state, or None when dense storage is as good.
    Rows are rounded up to 128 bytes when that costs at most 8 % (a frame whose rows are not whole cache lines has every row
    segment straddle lines and 64-byte write sectors: 64 x 8 x 1000 x 1000 runs the one-sweep Chambolle-Pock iteration in 9.2 ms
    padded against 18.1 ms dense, profiles/r4_pitch_bench.txt); otherwise -- short rows, where 128-byte rounding would add up to
    a quarter more bytes and measured SLOWER (256 x 4 x 100 x 100: 0.33 against 0.27 ms) -- only ragged rows (Nx not a multiple
    of the 16-byte lane) are rounded up to the lane, which takes them off the scalar-lane kernels.
    frame_pad_bytes: extra bytes between frames (round 4 measured that de-aliasing the frame pitch does NOT move the one-sweep
    kernel: tools/archive/bwtest4, tools/archive/alias_probe.py, profiles/r4_alias_probe.txt, profiles/r4_bwtest4_frame_pitch.txt; kept as an argument for experiments)."""
    es = 4 if dtype == torch.float32 else 8
    lane = 16 // es
    nx, ny = int(nx), int(ny)
    rp128 = ((nx * es + 127) // 128 * 128) // es
    if rp128 != nx and (rp128 - nx) <= 0.08 * nx:
        rp = rp128
    elif nx % lane != 0:
        rp = (nx + lane - 1) // lane * lane
    else:
        rp = nx
    pad = int(frame_pad_bytes) // es
    if rp == nx and pad == 0:
        return None
    return rp, ny * rp + pad


class _SlabProblem:
    """Common state: local slab geometry (and sub-slab geometries for interior / edge launches)."""

    def __init__(self, x0, scheme, reg_z_over_reg, reg_time, mask_static, factor_reg_static, slab, pitch=None):
        """pitch: "auto" (the solvers' default) = ``auto_pitch`` decides: dense for frames whose rows are whole cache lines (every
        BASELINE configuration), padded rows otherwise; None / "dense" = the state is dense like the caller's x0;
        (row_pitch, frame_pitch) in elements = every array of the solver's state (x0 is COPIED into such storage) is pitched
        (include/pytv4d.h, tv_geom::row_pitch / frame_pitch).  With padded state ``result()`` / ``x`` are (Nz, M, Ny, Nx) VIEWS of
        padded storage (``.contiguous()`` gives a dense copy)."""
        if not isinstance(x0, torch.Tensor) or not x0.is_cuda:
            raise ValueError("x0 must be a device tensor (this rank's z-slab of the volume)")
        if x0.dim() != 4:
            raise ValueError("x0 must be 4-D (Nz_local, M, N, N)")
        if isinstance(pitch, str):
            if pitch not in ("auto", "dense"):
                raise ValueError("pitch must be 'auto', 'dense' (or None) or (row_pitch, frame_pitch)")
            pitch = auto_pitch(x0.shape[2], x0.shape[3], x0.dtype) if pitch == "auto" else None
        self.pitch = (0, 0) if pitch is None else (int(pitch[0]), int(pitch[1]))
        self.x0 = x0.contiguous() if self.pitch == (0, 0) else x0
        self.device = x0.device
        self.dtype = x0.dtype
        self.scheme = scheme
        self.slab = slab if slab is not None else Slab(x0.shape[0], rank=0, world=1)
        if self.slab.nz != x0.shape[0]:
            raise ValueError("x0 has %d planes but this rank's slab has %d" % (x0.shape[0], self.slab.nz))
        self.kw = dict(reg_z_over_reg=reg_z_over_reg, reg_time=reg_time, mask_static=mask_static,
                       factor_reg_static=factor_reg_static)
        self._geoms = {}
        self.lib = _nv.lib()
        nz, m, ny, nx = self.x0.shape
        self._pkw = dict(row_pitch=self.pitch[0], frame_pitch=self.pitch[1])
        geo = _nv.Geometry((nz, m, ny, nx), scheme, self.dtype, self.device, nz_global=self.slab.nz_global, z0=self.slab.z0,
                           **self.kw, **self._pkw)
        if geo.pitched:
            x0p = geo.new_image()
            x0p.copy_(x0)
            self.x0 = x0p
        self._wvol = None
        if geo.weight_vol is not None:
            # per-voxel weights on the time regularisation (mask_static = float array of this rank's slab shape): the
            # neighbours' boundary planes of the weight are fetched ONCE (the sub-gradient's ghost-plane norms read them)
            wv = geo.weight_vol[0]
            gp = geo.new_image(1) if self.slab.prev is not None else None
            gn = geo.new_image(1) if self.slab.next is not None else None
            self.slab.wait(self.slab.exchange(send_prev=wv[0:1] if self.slab.prev is not None else None,
                                              send_next=wv[nz - 1:nz] if self.slab.next is not None else None,
                                              recv_prev=gp, recv_next=gn))
            self._wvol = (wv, gp, gn)
            twm = torch.tensor([geo.time_weight_max], dtype=torch.float64, device=self.device)
            if self.slab.sharded:
                self.slab.allreduce_max_(twm)
            self._twmax = float(twm.item())
            geo = None
        self.geo = geo if geo is not None else self.geom(0, nz)
        self._geoms[(0, nz)] = self.geo

    # ---- persistent small-volume loops (round 6; csrc/tv_small.hip) ------------------------------------------------------------------
    SMALL_MAX_VOXELS = 4 << 20      # the automatic rule: volumes of at most 4 Mvoxel take it (the reference's own shapes hold 0.07 - 1; measured against
                                    # the kernel pair / one-pass kernel from hipGraphs: x 2.5 - 3.7 up to 1 Mvoxel, x 1.5 - 2.2 at 4, x 1.0 - 1.1 at 9: profiles/r6_small_volume_larger.txt)
    SMALL_BLOCK = 128               # iterations per cooperative launch

    def _small_ok(self, persistent):
        """persistent: None = automatic (unsharded, <= SMALL_MAX_VOXELS, geometry supported), True = required, False = off"""
        if persistent is False:
            return False
        ok = (not self.slab.sharded) and bool(self.lib.tv_small_supported(self.geo.ref))
        if persistent is True:
            if not ok:
                raise ValueError("persistent=True: the persistent small-volume kernels take unsharded volumes that tv_small_supported accepts")
            return True
        return ok and self.x0.numel() <= self.SMALL_MAX_VOXELS

    def _small_ws(self):
        """scratch of the persistent kernels: block flags (zero-filled ONCE: their phase counters continue from launch to launch) + partials"""
        if getattr(self, "_small_ws_buf", None) is None:
            nbytes = self.lib.tv_small_workspace_bytes(self.geo.ref, self.SMALL_BLOCK)
            self._small_ws_buf = torch.zeros((nbytes + 7) // 8, dtype=torch.float64, device=self.device)
        return self._small_ws_buf

    def _small_check(self, loss):
        """the persistent kernels abandon a launch whose blocks cannot all run at once instead of hanging (csrc/tv_small.hip, small_sync): NaN history"""
        import numpy as _np
        if not _np.all(_np.isfinite(loss)):
            self._small_ws_buf = None
            raise RuntimeError("the persistent small-volume kernel returned a non-finite loss: either the iteration diverged or the launch was abandoned "
                               "(its blocks were not resident together); construct the solver with persistent=False or lower TV_SMALL_BLOCKS_PER_CU")
        return loss

    def geom(self, a, b):
        """Geometry of local planes [a, b) seen as a slab of the global volume."""
        key = (a, b)
        if key not in self._geoms:
            nz, m, ny, nx = self.x0.shape
            if self._wvol is None:
                self._geoms[key] = _nv.Geometry((b - a, m, ny, nx), self.scheme, self.dtype, self.device,
                                                nz_global=self.slab.nz_global, z0=self.slab.z0 + a, **self.kw, **self._pkw)
            else:
                wv, gp, gn = self._wvol
                kw = dict(self.kw, mask_static=False, factor_reg_static=0)
                dev = (wv[a:b], wv[a - 1] if a > 0 else (gp[0] if gp is not None else None),
                       wv[b] if b < nz else (gn[0] if gn is not None else None))
                g = _nv.Geometry((b - a, m, ny, nx), self.scheme, self.dtype, self.device, nz_global=self.slab.nz_global,
                                 z0=self.slab.z0 + a, weight_dev=dev, **kw, **self._pkw)
                g.time_weight_max = self._twmax
                self._geoms[key] = g
        return self._geoms[key]

    @property
    def stream(self):
        return _nv.current_stream(self.device)

    def new_plane(self, n=1):
        """n image planes (halo buffers) with the state's pitches"""
        _, m, ny, nx = self.x0.shape
        if self.pitch != (0, 0):
            return self.geo.new_image(n)
        return torch.empty((n, m, ny, nx), dtype=self.dtype, device=self.device)

    # ---- the arena (round 5, OPT-IN): ONE allocation for the arrays an iteration streams through ---------------------------------------
    # The time of the streaming kernels depends on where their arrays sit in physical memory (EXPERIMENTS.md section 3, round 4: the
    # "placement lottery" -- separate allocations land anywhere: 31.3 - 34.2 ms per north-star sweep from one construction to the next).
    # Carved out of ONE allocation with 24 - 48 MiB between consecutive arrays the sweep takes the SAME time construction after
    # construction (spread 0.4 - 0.5 % against 6 - 9 %: profiles/r5_slab_placement_probe{,2,3,4}.txt, tools/archive/slab_placement_probe.py) --
    # but WHICH time is decided by where the one big allocation lands: 31.1 ms on one box, 33.1 - 33.4 on three others, alternating
    # 31.05 / 32.1 from construction to construction on a fifth; a second arena measured beside the first (``_tune_arena``) does not
    # help where both land on the slow level (profiles/r5_bench_northstar_arena_first_command.json: 33.2 ms, 35.5 ms per iteration
    # against 33.8 with the per-array tuner).  Deterministic, not fast: it is therefore an OPTION (``arena=True``) for callers who
    # want run-to-run reproducible timing; the default stays separate allocations + the measuring tuner.
    ARENA_GAP_BYTES = 32 << 20
    _arena = None
    _arena_closed = False

    def _image_elems(self):
        nz, m, ny, nx = self.x0.shape
        return nz * m * (self.pitch[1] if self.pitch != (0, 0) else ny * nx)

    def _arena_open(self, n_images, n_grads, enable=None):
        """enable: None / False = off (the default), True = on; returns whether an arena is open"""
        img = self._image_elems()
        es = self.x0.element_size()
        if enable is None:
            enable = False
        if not enable or self.device.type != "cuda":
            return False
        gap = self.ARENA_GAP_BYTES // es
        al = 256 // es                            # every carve starts on a 256-byte boundary (round-5 advice: the C entry points want 16)
        rnd = lambda n: -(-n // al) * al
        total = n_images * (rnd(img) + gap) + n_grads * (rnd(img * self.geo.nd) + gap)
        try:
            self._arena = torch.empty(total, dtype=self.dtype, device=self.device)
        except RuntimeError:                      # no single block of that size: separate allocations (and the placement tuner)
            self._arena = None
            torch.cuda.empty_cache()
            return False
        self._arena_off, self._arena_gap, self._arena_align, self._arena_closed = 0, gap, al, False
        return True

    def _arena_close(self):
        """the state is carved: later new_image() / new_grad() calls allocate on their own"""
        self._arena_closed = True

    def _arena_take(self, elems):
        a = self._arena
        if a is None or self._arena_closed or self._arena_off + elems > a.numel():
            return None
        v = a[self._arena_off:self._arena_off + elems]
        self._arena_off += -(-elems // self._arena_align) * self._arena_align + self._arena_gap
        return v

    def new_image(self, zero=True):
        """an array like x0 (this rank's planes) with the state's pitches; pads always zero"""
        buf = self._arena_take(self._image_elems())
        if buf is not None:
            nz, m, ny, nx = self.x0.shape
            if self.pitch != (0, 0):
                buf.zero_()
                return buf.as_strided((nz, m, ny, nx), self.geo.image_strides())
            if zero:
                buf.zero_()
            return buf.view(nz, m, ny, nx)
        if self.pitch != (0, 0):
            return self.geo.new_image()
        return torch.zeros_like(self.x0) if zero else torch.empty_like(self.x0)

    def new_grad(self):
        """a zeroed gradient-like array (nz, Nd, M, Ny, Nx) with the state's pitches"""
        buf = self._arena_take(self._image_elems() * self.geo.nd)
        if buf is not None:
            nz, m, ny, nx = self.x0.shape
            nd = self.geo.nd
            buf.zero_()
            if self.pitch != (0, 0):
                fp, rp = self.geo.frame_pitch, self.geo.row_pitch
                return buf.as_strided((nz, nd, m, ny, nx), (nd * m * fp, m * fp, fp, rp, 1))
            return buf.view(nz, nd, m, ny, nx)
        return self.geo.new_grad()

    def image_copy(self, src):
        out = self.new_image(zero=False)
        out.copy_(src)
        return out


# =================================================================================================
class ChambollePock(_SlabProblem):
    """min_x 1/2 |x - x0|^2 + regularization * TV(x), README.md:141-157 with the state on the GPU.

    Per-step scalars: ``SLOTS`` fp64 device words, TV parts in [0:7], fidelity parts in [7:14] (one slot per
    launch; they are summed, over launches and over ranks, only when the loss is asked for).

    x0 : this rank's slab (device tensor).  ``step()`` enqueues one iteration; ``run(n)`` enqueues n
    and returns the loss history (one host synchronisation at the end)."""

    SLOTS = 14
    F = 7               # first fidelity slot

    @classmethod
    def loss_from_slots(cls, h, regularization):
        """README.md:157 loss from an (n, SLOTS) array of (already rank-summed) per-step scalars."""
        return h[:, cls.F:cls.SLOTS].sum(axis=1) + regularization * h[:, 0:cls.F].sum(axis=1)

    def __init__(self, x0, regularization, scheme="hybrid", reg_z_over_reg=1.0, reg_time=0.0, mask_static=False,
                 factor_reg_static=0, sigma_D=0.5, sigma_A=1.0, tau=None, slab=None, overlap=True, fused=None, pitch="auto",
                 q_pingpong=False, tune_placement=None, arena=None, persistent=None):
        """persistent (round 6): None = volumes of at most ``SMALL_MAX_VOXELS`` on one GPU run ``run`` / ``run_steps`` as ONE launch per
        ``SMALL_BLOCK`` iterations (tv_small_cp: the kernel pair's arithmetic inside a persistent kernel); False = never; True = required.
        fused: None = use the one-sweep kernel (tv_cp_fused + tv_cp_fixup: q read and written once per
        iteration) whenever the geometry supports it, False = always the dual + primal kernel pair.
        pitch: see ``_SlabProblem`` ("auto", the default: padded rows where that pays; None / "dense"; (row_pitch, frame_pitch)).
        q_pingpong (one-sweep path; default off): read the dual variable from one array and write it to a second one, swapping them
        every iteration, instead of updating it in place.  An arithmetic-free kernel with the sweep's memory shape gains 9 % from
        it (tools/archive/bwtest4, profiles/r4_bwtest4_q_pingpong.txt); the real sweep does not (tools/archive/pp_probe.py: 33.3 against 33.4 ms in
        one pool, and one of the two directions can be 3 ms slower than the other when the arrays are separate allocations) --
        kept as an option of tv_cp_sweep, not used by default.
        arena (round 5, default off): True = x, x_alt, p, a private copy of x0 and q are carved out of ONE allocation with 32 MiB
        between them (``_SlabProblem._arena_open``): the sweep then takes the same time in every construction (spread 0.5 % instead
        of 6 - 9 %) -- on whatever level that allocation landed, which is the slow one more often than not; see the comment there.
        tune_placement: None = on for volumes (slabs) of >= 4 GiB per image with memory to spare and no arena (see
        ``_tune_x_placement``)."""
        super().__init__(x0, scheme, reg_z_over_reg, reg_time, mask_static, factor_reg_static, slab, pitch=pitch)
        self.reg = float(regularization)
        self.sigma_D, self.sigma_A = float(sigma_D), float(sigma_A)
        self.tau = float(tau) if tau is not None else cp_step_size(self.slab.nz_global, x0.shape[1], reg_z_over_reg, reg_time,
                                                                    self.geo.time_weight_max)
        fused_given = fused is not None
        if fused is None and persistent is True:
            fused = False                                 # the persistent loop was asked for: it replaces the kernel pair, not the one-sweep kernel
        if fused is None:
            # one-sweep kernel where supported -- except on volumes too small to fill the GPU with its blocks (8 rows x 256
            # columns x >= 8 planes x all frames each: a block holds >= 16 k x M voxels and a CU takes 8 / M of them, so
            # 256 CUs want >= 33 Mvoxel for one full round; tools/archive/fused_vs_pair.py: 0.5 - 0.7 x of the kernel pair at
            # 6 - 8 Mvoxel with few planes, break-even at 8 - 13 Mvoxel, 1.15 - 1.25 x faster from 16 Mvoxel on whatever
            # the plane size -- 256x1x512x512 runs 990 - 1040 it/s against 870).  Option TV_FUSED_MIN_KVOXELS (thousands of
            # voxels of this rank's slab) moves the switch.
            min_vox = 1024 * _nv.get_option("TV_FUSED_MIN_KVOXELS", 16384)
            fused = bool(self.lib.tv_cp_fused_supported(self.geo.ref)) and self.x0.numel() >= min_vox
        # (an explicit fused=False asks for the kernel pair: the persistent loop replaces it only when it is asked for as well)
        if bool(fused) and persistent is True:
            raise ValueError("fused=True and persistent=True are two different kernels: ask for one")
        self.small = (not bool(fused)) and self._small_ok(False if (fused_given and persistent is None) else persistent)
        self.fused = bool(fused)
        if self.fused and not self.lib.tv_cp_fused_supported(self.geo.ref):
            raise ValueError("the one-sweep Chambolle-Pock kernel does not support this geometry (tv_cp_fused_supported: "
                             "fp32, Nx % 4 == 0, Nx >= 64, Ny * Nx <= 2^30; any number of frames)")
        # the arrays of the iteration, in the order the arena was measured with: x, x_alt, p, x0 (the solver's own copy), q
        self._x0_src, self._q_pingpong, self._arena_want = self.x0, bool(q_pingpong), arena
        self.arena = False
        self._alloc_state()
        self.ws = self.geo.workspace()
        self.plan = HaloPlan(self.slab, scheme, self.geo.z_active)
        pl = self.plan
        sh = pl.on
        self.ch_back, self.ch_fwd = pl.ch_back, pl.ch_fwd
        self.xh_prev = self.new_plane() if pl.x_need_prev else None
        self.xh_next = self.new_plane() if pl.x_need_next else None
        self.qh_prev = self.new_plane() if pl.g_need_prev else None
        self.qh_next = self.new_plane() if pl.g_need_next else None
        self.overlap = bool(overlap) and sh and self.slab.nz >= 3 and not self.fused
        self.hist = None
        self.it = 0
        self.timing = None      # set to a list to collect (start, after kernel 1, after kernel 2) HIP events per step
        self.phase_timing = None  # set to a list to collect one (name, event) list per step: every phase boundary of the schedule
        self._scratch = torch.zeros(self.SLOTS, dtype=torch.float64, device=self.device)
        self._lag = None                 # None: every step returns its own fidelity; else: lagged-fidelity block (see run_steps)
        self._final = False              # the block's last iteration: its sweep returns BOTH fidelities (TV_CP_FID_BOTH), its fix-up reads x0
        self._lag_void = torch.zeros(1, dtype=torch.float64, device=self.device)
        self._both = torch.zeros((self.SLOTS, 2), dtype=torch.float64, device=self.device)     # [slot][input, output] of the final sweeps
        self._cur_out = self._scratch
        if self.fused:
            self.zchunk = int(self.lib.tv_cp_zchunk(self.geo.ref))
            self.nchunks = (self.slab.nz + self.zchunk - 1) // self.zchunk
            # interior-first scheduling of the one-sweep path needs an interior: >= 3 chunks and >= 3 planes
            self.overlap_fused = bool(overlap) and sh and self.nchunks >= 3 and self.slab.nz >= 3
        self.placement = None
        if self.fused and self.arena and tune_placement is None:
            # an arena of >= 4 GiB images with room for a second one: measure two, keep the faster (never on a sharded slab's behalf of others:
            # local sweeps only)
            img_bytes = self.x.numel() * self.x.element_size()
            free, _total = torch.cuda.mem_get_info(self.device)
            if img_bytes >= (4 << 30) and free >= self._arena.numel() * self._arena.element_size() + (8 << 30):
                self._tune_arena()
        if self.fused:
            img_bytes = self.x.numel() * self.x.element_size()
            if tune_placement is None:
                free, _total = torch.cuda.mem_get_info(self.device)
                # (sharded slabs too: the tuner launches local kernels only, no rank waits for another; every rank of a weak-scaling run
                # holds a slab of the single-GPU size and plays the same placement lottery)
                tune_placement = (not self.arena) and img_bytes >= (4 << 30) and free >= 3 * img_bytes + (8 << 30)
            if tune_placement:
                # (q is NOT pinned here: the tuner keeps exactly one q bound at any time -- the best so far -- and at most one
                # candidate beside it, so that a winning candidate frees the original before the next one is allocated; round-4 advice)
                keep = (self.x, self.x_alt, self.p, self.x0)
                try:
                    self._tune_x_placement()
                except RuntimeError as exc:       # out of memory while holding the candidates, ...: the tuner is an optimisation, never a failure
                    self.x, self.x_alt, self.p, self.x0 = keep
                    self._lag = None
                    self.x.copy_(self.x0)
                    self.p.zero_()
                    self.q.zero_()
                    torch.cuda.empty_cache()
                    self.placement = {"error": str(exc)[:200]}
                del keep

    def _alloc_state(self):
        """(Re)allocate x, x_alt, p, the private copy of x0, q [, q_alt] -- out of a fresh arena when one can be had -- and put them
        into the initial state (x = x0, p = q = 0)."""
        self._arena = None
        want = self._arena_want
        self.arena = self._arena_open(4 if self.fused else 3, 2 if (self.fused and self._q_pingpong) else 1, want) if (self.fused or want) else False
        self.x0 = self._x0_src
        self.x = self.image_copy(self.x0)
        self.x_alt = self.new_image() if self.fused else None      # ping-pong partner of x
        self.p = self.new_image()
        if self.arena:
            self.x0 = self.image_copy(self._x0_src)
        self.q = self.new_grad()
        self.q_alt = self.new_grad() if (self.fused and self._q_pingpong) else None
        if self.arena:
            self._arena_close()

    def _tune_arena(self, reps=2):
        """Two arenas, measured with the real sweep, the faster one kept (round 5).  WHERE one big allocation lands decides the level
        of every sweep that runs on it -- 31.0 or 32.1 ms on one box, 31.2 or 33.1 on others, the same for every construction that gets
        the same region and independent of the gap between the arrays (profiles/r5_slab_placement_probe{3,4}.txt) -- and two
        allocations made one after the other land in different regions.  Costs one more arena for a moment and ~8 sweeps; the state
        is re-initialised afterwards, results do not depend on it."""
        import time as _time
        t_begin = _time.perf_counter()
        out = torch.zeros(self.SLOTS, dtype=torch.float64, device=self.device)

        def round_trip():
            hp = self.x[0:1] if self.plan.x_need_prev else None
            hn = self.x[0:1] if self.plan.x_need_next else None
            ts = []
            for r in range(reps + 1):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                self._sweep(0, -1, hp, hn, out[0:1], out[self.F:self.F + 1])
                self.x, self.x_alt = self.x_alt, self.x
                self._sweep(0, -1, hp, hn, out[0:1], out[self.F:self.F + 1])
                self.x, self.x_alt = self.x_alt, self.x
                b.record()
                ts.append((a, b))
            torch.cuda.synchronize(self.device)
            return min(a.elapsed_time(b) for a, b in ts[1:])

        names = ("_arena", "_arena_off", "_arena_gap", "x", "x_alt", "p", "x0", "q", "q_alt")
        t0 = round_trip()
        first = {k: getattr(self, k) for k in names}
        info = {"round_trip_ms": [round(t0, 3)]}
        try:
            self._alloc_state()
            if not self.arena:
                raise RuntimeError("no second arena")
            t1 = round_trip()
            info["round_trip_ms"].append(round(t1, 3))
            if t0 <= t1:                         # keep the first: the second loses its last references here
                for k, v in first.items():
                    setattr(self, k, v)
        except RuntimeError as exc:              # out of memory for a second arena, ...: an optimisation, never a failure
            for k, v in first.items():
                setattr(self, k, v)
            self.arena = True
            info["error"] = str(exc)[:160]
        del first
        torch.cuda.empty_cache()
        self.x.copy_(self.x0)
        self.p.zero_()
        self.q.zero_()
        if self.q_alt is not None:
            self.q_alt.zero_()
        torch.cuda.synchronize(self.device)
        info["chosen"] = int(len(info["round_trip_ms"]) == 2 and info["round_trip_ms"][1] < info["round_trip_ms"][0])
        info["seconds"] = round(_time.perf_counter() - t_begin, 3)
        self.placement = {"arena": info}

    # ---- one phase on local planes [a, b) -----------------------------------------------------
    def _dual(self, a, b, xp, xn, out):
        g = self.geom(a, b)
        _nv.check(self.lib.tv_cp_dual(g.ref, _nv.ptr(self.x[a:b]), _nv.ptr(xp), _nv.ptr(xn), _nv.ptr(self.q[a:b]),
                                      self.sigma_D, self.reg, out.data_ptr(), _nv.ptr(self.ws), self.stream))

    def _primal(self, a, b, qp, qn, out):
        g = self.geom(a, b)
        _nv.check(self.lib.tv_cp_primal(g.ref, _nv.ptr(self.q[a:b]), _nv.ptr(qp), _nv.ptr(qn), _nv.ptr(self.x[a:b]),
                                        _nv.ptr(self.x0[a:b]), _nv.ptr(self.p[a:b]), self.tau, self.sigma_A,
                                        out.data_ptr(), _nv.ptr(self.ws), self.stream))

    def _tune_x_placement(self, n_extra=2, reps=2):
        """Pick WHERE the arrays of the iteration live by MEASUREMENT: the dual variable q (one alternative allocation, when the
        memory is there), the two image buffers the iterate ping-pongs between (``n_extra`` more candidates, every ordered pair
        timed), and the fidelity dual p (the candidates that are left).

        Why: on MI355X the time of the one-sweep kernel depends on where its arrays landed in
        physical memory -- the same binary on the same data runs the north-star sweep in 31.2 or in 34 ms, the level is fixed for
        the life of an allocation, and it differs between the two directions of the x ping-pong (sweeps alternate 34.5 / 32.5 ms):
        the memory side answers the reads of one placement ~4 % later than those of another (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ,
        evenly over all 128 L2 channels: profiles/r4_channel_counters.txt; the rate of a plain copy depends on the distance
        between source and destination in the same way: profiles/r4_deltatest.txt).  Nothing a process can ask the allocator for
        controls it (page size, alignment, one pool or many, a pitch: profiles/r4_vmtest.txt, r4_bwtest4_frame_pitch.txt,
        r3_placement_experiments.txt), but it can be MEASURED.  Costs ~40 sweeps once per solver (~1.3 s for the north star:
        same-box A/B 33.8 / 34.3 / 34.8 ms per iteration with it, 35.2 / 35.4 / 35.2 without, profiles/r4_placement_tuner_ab.txt);
        the state is re-initialised afterwards, results do not depend on it."""
        import time as _time
        t_begin = _time.perf_counter()
        out = torch.zeros(self.SLOTS, dtype=torch.float64, device=self.device)
        hp = self.x[0:1] if self.plan.x_need_prev else None           # timing only: any plane of the right layout serves as a halo
        hn = self.x[0:1] if self.plan.x_need_next else None
        info = {}

        def one(i_buf, o_buf):
            self.x, self.x_alt = i_buf, o_buf
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self._sweep(0, -1, hp, hn, out[0:1], out[self.F:self.F + 1])
            b.record()
            return a, b

        def round_trip(u, v):
            """ms of u -> v plus v -> u (best of ``reps``)"""
            evs = [(one(u, v), one(v, u)) for _ in range(reps)]
            torch.cuda.synchronize(self.device)
            return min(e[0][0].elapsed_time(e[0][1]) for e in evs) + min(e[1][0].elapsed_time(e[1][1]) for e in evs)

        x_a, x_b = self.x, self.x_alt
        for _ in range(2):                                                   # warm-up (code load, clocks)
            one(x_a, x_b)
        # ---- the dual variable: one alternative allocation, if it fits with room to spare ---------------------------------------
        q_bytes = self.q.numel() * self.q.element_size()
        img_bytes = x_a.numel() * x_a.element_size()
        free, _total = torch.cuda.mem_get_info(self.device)
        if self.q_alt is None and free >= q_bytes + (n_extra + 1) * img_bytes + (8 << 30):
            # up to two alternatives, one at a time (never more than two q arrays alive): a slow q allocation costs up to 2 ms per
            # sweep (profiles/r4_bench_northstar_first_command_e.json: round trips 67.3 vs 63.5 ms)
            best_q, best_t = self.q, round_trip(x_a, x_b)
            info["q_round_trip_ms"] = [round(best_t, 3)]
            try:
                for _ in range(2):
                    cand = self.new_grad()
                    self.q = cand
                    t2 = round_trip(x_a, x_b)
                    info["q_round_trip_ms"].append(round(t2, 3))
                    if t2 < best_t:
                        best_q, best_t = cand, t2
                    self.q = best_q                  # the loser (the original included) loses its last reference here
                    del cand
                    torch.cuda.empty_cache()
            except RuntimeError as exc:              # no room for another candidate: keep the best q measured so far, go on with the images
                self.q = best_q
                torch.cuda.empty_cache()
                info["q_error"] = str(exc)[:160]
            del best_q
        # ---- the image pair: every ordered pair of the candidates ------------------------------------------------------------------
        cands = [x_a, x_b] + [self.new_image() for _ in range(n_extra)]
        n = len(cands)
        ev = [[[] for _ in range(n)] for _ in range(n)]
        for r in range(reps):
            for i in range(n):
                for j in range(n):
                    if i != j:
                        ev[i][j].append(one(cands[i], cands[j]))
        torch.cuda.synchronize(self.device)
        t = [[(min(a.elapsed_time(b) for a, b in ev[i][j]) if i != j else float("inf")) for j in range(n)] for i in range(n)]
        _, bi, bj = min(((t[i][j] + t[j][i], i, j) for i in range(n) for j in range(i + 1, n)))
        info.update({"candidates": n, "sweep_ms": [[None if i == j else round(t[i][j], 3) for j in range(n)] for i in range(n)],
                     "chosen": [bi, bj], "chosen_ms": [round(t[bi][bj], 3), round(t[bj][bi], 3)],
                     "first_pair_ms": [round(t[0][1], 3), round(t[1][0], 3)]})
        x_a, x_b = cands[bi], cands[bj]
        # ---- the fidelity dual p: the candidates that are left, against the allocation it has ------------------------------------------
        rest = [c for k, c in enumerate(cands) if k not in (bi, bj)]
        del cands
        p_cands = [self.p] + rest
        tp = []
        for c in p_cands:
            self.p = c
            tp.append(round_trip(x_a, x_b))
        kp = min(range(len(tp)), key=lambda k: tp[k])
        self.p = p_cands[kp]
        info["p_round_trip_ms"] = [round(v, 3) for v in tp]
        info["p_chosen"] = kp
        # ---- x0 (read once per sweep): the caller's array against a copy in one of the buffers that are left ------------------------
        spare = [c for k, c in enumerate(p_cands) if k != kp]
        del p_cands, rest
        if spare:
            x0_orig, x0_copy = self.x0, spare[0]
            x0_copy.copy_(x0_orig)
            t0_ = round_trip(x_a, x_b)
            self.x0 = x0_copy
            t1_ = round_trip(x_a, x_b)
            info["x0_round_trip_ms"] = [round(t0_, 3), round(t1_, 3)]
            if t0_ <= t1_:
                self.x0 = x0_orig
            del x0_orig, x0_copy
        del spare
        self.x, self.x_alt = x_a, x_b
        # back to the initial state: x = x0, p = q = 0 (the timed sweeps wrote into them)
        self.x.copy_(self.x0)
        self.p.zero_()
        self.q.zero_()
        if self.q_alt is not None:
            self.q_alt.zero_()
        torch.cuda.synchronize(self.device)
        torch.cuda.empty_cache()
        info["seconds"] = round(_time.perf_counter() - t_begin, 3)
        self.placement = info

    def _sweep(self, c0, cn, xp, xn, tv_slot, fid_slot):
        """One sweep launch.  In a lagged-fidelity block (``_lag``, see ``_run_eager``) the sweep returns 1/2 |x_in - x0|^2 over all
        sites -- the fidelity of the iterate the PREVIOUS step produced -- into the previous step's slot, and the fix-up reads no x0."""
        g = self.geo
        flags = 0
        own_slot, both = fid_slot, None
        if self._lag is not None:
            flags = 1                                    # TV_CP_FID_OF_INPUT
            i = fid_slot.storage_offset() - self._cur_out.storage_offset()
            if self._lag is False:
                fid_slot = self._lag_void                # first sweep of a block: nobody waits for its input's fidelity
            else:                                        # the same slot index, one row back
                fid_slot = self._lag[i:i + 1]
            if self._final:
                # last iteration of the block (round 5): the sweep returns the fidelity of its input AND of its output (complete sites), the
                # fix-up -- called with x0 -- the rest: no reduction pass over x and x0 closes the block any more
                flags = 3                                # TV_CP_FID_OF_INPUT | TV_CP_FID_BOTH
                both, prev_slot = self._both[i], fid_slot
                fid_slot = both
        _nv.check(self.lib.tv_cp_sweep(g.ref, _nv.ptr(self.x), _nv.ptr(xp), _nv.ptr(xn), _nv.ptr(self.q),
                                       _nv.ptr(self.q_alt if self.q_alt is not None else self.q), _nv.ptr(self.x0),
                                       _nv.ptr(self.p), _nv.ptr(self.x_alt), self.sigma_D, self.reg, self.tau, self.sigma_A, flags,
                                       c0, cn, tv_slot.data_ptr(), fid_slot.data_ptr(), _nv.ptr(self.ws), self.stream))
        if both is not None:
            prev_slot.copy_(both[0:1])
            own_slot.copy_(both[1:2])

    def _fixup(self, z0, zn, qp, qn, fid_slot):
        g = self.geo
        lag = self._lag is not None and not self._final
        _nv.check(self.lib.tv_cp_fixup(g.ref, _nv.ptr(self.q_alt if self.q_alt is not None else self.q), _nv.ptr(qp), _nv.ptr(qn), _nv.ptr(self.x_alt),
                                       None if lag else _nv.ptr(self.x0), self.tau, z0, zn,
                                       (self._lag_void if lag else fid_slot).data_ptr(), _nv.ptr(self.ws), self.stream))

    def _step_fused(self, out):
        """One-sweep iteration: x halos -> sweep (x -> x_alt) -> q' halos -> fix-up -> swap.  With a sharded
        slab the interior chunks / planes run while the halo planes are in flight: the first half of the interior chunks
        hides the x exchange, then the two edge chunks run, and the SECOND half of the interior chunks (with the interior
        fix-up behind it) hides the q' exchange -- the interior fix-up alone is too short for that on a 32-plane slab.
        ``phase_timing`` (a list): one event per phase boundary on the launch stream, so that a scaling run can say where an
        iteration's time went -- in particular how long the stream sat in ``s.wait(h)`` with nothing left to overlap."""
        s, nz, F = self.slab, self.slab.nz, self.F
        self._cur_out = out
        ev = self._events()
        mark = self._phase_marker()
        qhp = self.qh_prev[0] if self.qh_prev is not None else None
        qhn = self.qh_next[0] if self.qh_next is not None else None
        mark("start")
        h = self.plan.exchange_image(self.x, self.xh_prev, self.xh_next)
        if ev:
            ev[0].record()
        if self.overlap_fused:
            nch = self.nchunks
            na = (nch - 2 + 1) // 2                      # interior chunks before the edge chunks, nb after them
            nb = nch - 2 - na
            self._sweep(1, na, None, None, out[0:1], out[F:F + 1])
            mark("sweep_interior_a")
            s.wait(h)
            mark("x_halo_wait_exposed")
            self._sweep(0, 1, self.xh_prev, None, out[1:2], out[F + 1:F + 2])
            self._sweep(nch - 1, 1, None, self.xh_next, out[2:3], out[F + 2:F + 3])
            mark("sweep_edges")
            h = self.plan.exchange_grad(self.q_alt if self.q_alt is not None else self.q, qhp, qhn)
            if nb > 0:
                self._sweep(1 + na, nb, None, None, out[3:4], out[F + 3:F + 4])
            mark("sweep_interior_b")
            if ev:
                ev[1].record()
        else:
            s.wait(h)
            mark("x_halo_wait_exposed")
            self._sweep(0, -1, self.xh_prev, self.xh_next, out[0:1], out[F:F + 1])
            mark("sweep")
            if ev:
                ev[1].record()
            h = self.plan.exchange_grad(self.q_alt if self.q_alt is not None else self.q, qhp, qhn)
        if self.overlap_fused:
            self._fixup(1, nz - 2, None, None, out[F + 4:F + 5])
            mark("fixup_interior")
            s.wait(h)
            mark("q_halo_wait_exposed")
            self._fixup(0, 1, self.qh_prev, None, out[F + 5:F + 6])
            self._fixup(nz - 1, 1, None, self.qh_next, out[F + 6:F + 7])
            mark("fixup_edges")
        else:
            s.wait(h)
            mark("q_halo_wait_exposed")
            self._fixup(0, -1, self.qh_prev, self.qh_next, out[F + 4:F + 5])
            mark("fixup")
        if ev:
            ev[2].record()
        self.x, self.x_alt = self.x_alt, self.x
        if self.q_alt is not None:
            self.q, self.q_alt = self.q_alt, self.q          # self.q is always the current dual variable
        self.it += 1
        if self._lag is not None:
            self._lag = out                      # the next sweep delivers THIS step's fidelity into these slots

    def _phase_marker(self):
        """mark(name): record an event on the launch stream that closes phase `name` (no-op unless ``phase_timing`` is a list)"""
        if self.phase_timing is None:
            return lambda name: None
        rec = []
        self.phase_timing.append(rec)

        def mark(name):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            rec.append((name, e))
        return mark

    @staticmethod
    def phase_means_ms(phase_timing):
        """{phase: mean milliseconds per step} from the event lists collected in ``phase_timing`` (synchronise first)"""
        acc, order = {}, []
        for rec in phase_timing:
            for (_, a), (name, b) in zip(rec[:-1], rec[1:]):
                if name not in acc:
                    acc[name] = []
                    order.append(name)
                acc[name].append(a.elapsed_time(b))
        return {k: float(sum(acc[k]) / len(acc[k])) for k in order}

    def _events(self):
        if self.timing is None:
            return None
        ev = tuple(torch.cuda.Event(enable_timing=True) for _ in range(3))
        self.timing.append(ev)
        return ev

    def step(self, out=None):
        """Enqueue one iteration.  out: fp64 device tensor of SLOTS words receiving this rank's TV parts
        [0:7] and fidelity parts [7:14] (summed later); defaults to an internal scratch."""
        out = self._scratch if out is None else out
        if self.fused:
            return self._step_fused(out)
        nz, s, F = self.slab.nz, self.slab, self.F
        x, q = self.x, self.q
        ev = self._events()
        mark = self._phase_marker()
        mark("start")
        # ---------------- dual: q <- proj(q + sigma D x) -----------------------------------------
        h = self.plan.exchange_image(x, self.xh_prev, self.xh_next)
        if ev:
            ev[0].record()
        if self.overlap:
            self._dual(1, nz - 1, x[0:1], x[nz - 1:nz], out[0:1])
            mark("dual_interior")
            s.wait(h)
            mark("x_halo_wait_exposed")
            self._dual(0, 1, self.xh_prev, x[1:2], out[1:2])
            self._dual(nz - 1, nz, x[nz - 2:nz - 1], self.xh_next, out[2:3])
            mark("dual_edges")
        else:
            s.wait(h)
            mark("x_halo_wait_exposed")
            self._dual(0, nz, self.xh_prev, self.xh_next, out[0:1])
            mark("dual")
        if ev:
            ev[1].record()
        # ---------------- primal: x <- x - tau p - tau D^T q ---------------------------------------
        h = self.plan.exchange_grad(q, self.qh_prev[0] if self.qh_prev is not None else None,
                                    self.qh_next[0] if self.qh_next is not None else None)
        if self.overlap:
            self._primal(1, nz - 1, q[0, self.ch_back], q[nz - 1, self.ch_fwd], out[F:F + 1])
            mark("primal_interior")
            s.wait(h)
            mark("q_halo_wait_exposed")
            self._primal(0, 1, self.qh_prev, q[1, self.ch_fwd], out[F + 1:F + 2])
            self._primal(nz - 1, nz, q[nz - 2, self.ch_back], self.qh_next, out[F + 2:F + 3])
            mark("primal_edges")
        else:
            s.wait(h)
            mark("q_halo_wait_exposed")
            self._primal(0, nz, self.qh_prev, self.qh_next, out[F:F + 1])
            mark("primal")
        if ev:
            ev[2].record()
        self.it += 1

    GRAPH_BLOCK = 10            # iterations captured per hipGraph (even: the x ping-pong returns to its start)
    GRAPH_MAX_VOXELS = 1 << 23  # below this an iteration is launch-bound (tens of microseconds of kernels)

    def run(self, n_iter, record_loss=True, graph=None):
        """n_iter iterations; returns the README's loss history (README.md:157) as a numpy array
        (global over all ranks), or None.  graph: None = capture the loop in a hipGraph when the problem is
        small enough to be launch-bound and not sharded; True / False force it."""
        hist = torch.zeros((n_iter, self.SLOTS), dtype=torch.float64, device=self.device)
        use_graph = (self.x0.numel() <= self.GRAPH_MAX_VOXELS and not self.slab.sharded and self.timing is None and self.phase_timing is None) if graph is None else bool(graph)
        if self._small_now() and graph is None:
            use_graph = False                     # the persistent kernel IS the loop: nothing left to capture
        start = 0
        if use_graph and n_iter >= 2 + 2 * self.GRAPH_BLOCK and not self.slab.sharded:
            # two eager iterations (also the warm-up of the capture), then graph replays, then an eager tail
            self.step(hist[0])
            self.step(hist[1])
            done = self._run_graphed_from(hist, 2, n_iter)
            start = 2 + max(done, 0)
        self.run_steps(hist[start:n_iter])
        if not record_loss:
            return None
        self.slab.allreduce_sum_(hist)
        loss = self.loss_from_slots(hist.cpu().numpy(), self.reg)
        return self._small_check(loss) if self.small else loss

    def run_steps(self, rows):
        """Enqueue ``len(rows)`` iterations, row k of the (n, SLOTS) fp64 device tensor ``rows`` receiving the scalars of iteration k.
        One-sweep path (round 4): the fidelity 1/2 |x_{k+1} - x0|^2 of row k is delivered by the sweep of iteration k+1 (it has
        x_{k+1} and x0 in registers: tv_cp_sweep, TV_CP_FID_OF_INPUT), so the fix-up reads no x0; the LAST iteration's sweep returns the
        fidelity of its output as well (TV_CP_FID_BOTH, round 5: over the complete sites; its fix-up, called with x0, adds the rest).
        The rows must be zero on entry (slots are written once each)."""
        n = rows.shape[0]
        if n == 0:
            return
        if self._small_now():
            return self._run_small(rows)
        if not self.fused:
            for k in range(n):
                self.step(rows[k])
            return
        self._lag = False                # first sweep of the block: its input's fidelity belongs to nobody
        try:
            for k in range(n):
                self._final = (k == n - 1)       # the last sweep returns both fidelities (round 4: one more pass over x and x0 closed the block)
                self.step(rows[k])
        finally:
            self._lag = None
            self._final = False

    def _small_now(self):
        """the persistent path runs the loop unless somebody asked for per-step events (bench.py: ``timing`` / ``phase_timing``)"""
        return self.small and self.timing is None and self.phase_timing is None

    def _run_small(self, rows):
        """``len(rows)`` iterations in blocks of ``SMALL_BLOCK`` per cooperative launch (tv_small_cp): x, p, q updated in place; TV of the
        iterate each dual update saw -> slot 0, 1/2 |x_new - x0|^2 -> slot F (the slots the kernel pair fills)."""
        ws = self._small_ws()
        n, done = rows.shape[0], 0
        if rows.stride(1) != 1:
            raise ValueError("rows must be a (n, SLOTS) fp64 tensor with contiguous rows")
        while done < n:
            k = min(self.SMALL_BLOCK, n - done)
            # the reduction that closes the launch writes TV -> slot 0 and the fidelity -> slot F of the rows themselves
            _nv.check(self.lib.tv_small_cp(self.geo.ref, _nv.ptr(self.x), _nv.ptr(self.x0), _nv.ptr(self.p), _nv.ptr(self.q), self.sigma_D, self.reg,
                                           self.tau, self.sigma_A, k, rows[done].data_ptr(), rows.stride(0), self.F, _nv.ptr(ws), self.stream))
            done += k
            self.it += k

    def _run_graphed_from(self, hist, first, n_iter):
        """Capture GRAPH_BLOCK iterations (not executed during capture) and replay them over hist[first:]."""
        K = self.GRAPH_BLOCK
        nrep = (n_iter - first) // K
        if nrep < 1:
            return 0
        it0, x_ref, xalt_ref, q_ref, qalt_ref = self.it, self.x, self.x_alt, self.q, self.q_alt
        try:
            buf = torch.zeros((K, self.SLOTS), dtype=torch.float64, device=self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for k in range(K):
                    self.step(buf[k])
        except Exception:
            self.it, self.x, self.x_alt, self.q, self.q_alt = it0, x_ref, xalt_ref, q_ref, qalt_ref      # nothing ran: undo the bookkeeping, stay eager
            return 0
        self.it = it0
        done = 0
        for r in range(nrep):
            graph.replay()
            hist[first + done:first + done + K].copy_(buf)
            done += K
            self.it += K
        return done

    def result(self):
        """The iterate as a DENSE (Nz, M, Ny, Nx) tensor: with padded state (``pitch``) a contiguous copy -- callers that ``.view(-1)``
        it, take its ``data_ptr()`` or hand it to the dense ``tv_*`` entry points must not see pad columns (round-4 advice); ``.x``
        stays the raw (possibly strided) view of the solver's storage."""
        return self.x.contiguous() if self.pitch != (0, 0) else self.x


# =================================================================================================
class ChambollePockOperator(_SlabProblem):
    """min_x 1/2 |A x - b|^2 + regularization * TV(x) for a user-supplied linear operator A (the CT use case the
    reference is written for: README.md:2; its CP snippet is the special case A = I, README.md:148 ``sigma_A``).

        p <- (p + sigma_A (A x - b)) / (1 + sigma_A)
        q <- proj_{|.|_2 <= reg}(q + sigma_D D x)                       HIP: ONE sweep over q for both lines (tv_cpop_fused +
        x <- x - tau A^T p - tau D^T q                                   tv_cpop_fixup, round 3) or tv_cp_dual + tv_DT_axpy2

    ``A`` and ``AT`` are callables taking and returning DEVICE tensors (``A``: image (Nz, M, Ny, Nx) -> data of any
    shape, ``AT``: data -> image); they stay the user's code, the TV part runs in the HIP kernels.  ``A`` is applied ONCE by the
    constructor (the residual A x_init - b is carried from iteration to iteration); ``b`` must live on x_init's GPU.  ``tau`` must
    satisfy tau (sigma_A |A|^2 + sigma_D |D|^2) <= 1; the default assumes |A| <= 1.

    With a ``slab`` (one process per GPU) ``x_init`` / ``b`` are this rank's z-slab of the image and its share of the data,
    ``A`` / ``AT`` act on the slab (operators that couple the slabs -- a cone-beam projector, say -- do their own
    communication inside the callables); the TV part trades the image / gradient halo planes like ``ChambollePock``."""

    def __init__(self, A, AT, b, x_init, regularization, scheme="hybrid", reg_z_over_reg=1.0, reg_time=0.0,
                 mask_static=False, factor_reg_static=0, sigma_D=0.5, sigma_A=1.0, tau=None, slab=None, fused=None):
        super().__init__(x_init, scheme, reg_z_over_reg, reg_time, mask_static, factor_reg_static, slab)
        if not isinstance(b, torch.Tensor) or not b.is_cuda or b.device != self.device:
            # b's raw pointer goes to HIP kernels (tv_cpop_residual, tv_cpop_p): a host tensor or one on another GPU would be a
            # memory fault there, not the device-mismatch error torch used to raise (round-3 advice)
            raise ValueError("b must be a device tensor on the same GPU as x_init (%s), got %s"
                             % (self.device, b.device if isinstance(b, torch.Tensor) else type(b).__name__))
        self.A, self.AT = A, AT
        # fused (round 3): the TV part as ONE sweep over q (tv_cpop_fused + tv_cpop_fixup: the one-sweep Chambolle-Pock kernel with
        # A^T p in the place of the fidelity dual) instead of tv_cp_dual + tv_DT_axpy2 -- 2 Nd + 4 words per voxel instead of 3 Nd + 4
        can_fuse = bool(self.lib.tv_cp_fused_supported(self.geo.ref))
        if fused and not can_fuse:
            raise ValueError("fused=True needs a geometry tv_cp_fused_supported() accepts")
        self.fused = can_fuse if fused is None else bool(fused)
        self.reg = float(regularization)
        self.sigma_D, self.sigma_A = float(sigma_D), float(sigma_A)
        self.tau = float(tau) if tau is not None else cp_step_size(self.slab.nz_global, x_init.shape[1], reg_z_over_reg, reg_time,
                                                                    self.geo.time_weight_max)
        self.x = self.x0.clone()
        self.b = b.to(self.dtype).contiguous()
        self.p = torch.zeros_like(self.b)
        self.q = torch.zeros(self.geo.grad_shape, dtype=self.dtype, device=self.device)
        self.ws = self.geo.workspace()
        self.x_new = torch.empty_like(self.x)
        self.plan = HaloPlan(self.slab, scheme, self.geo.z_active)
        pl = self.plan
        self.xh_prev = self.new_plane() if pl.x_need_prev else None
        self.xh_next = self.new_plane() if pl.x_need_next else None
        self.qh_prev = self.new_plane() if pl.g_need_prev else None
        self.qh_next = self.new_plane() if pl.g_need_next else None
        # the residual r = A x - b is CARRIED from one iteration to the next (round 2 applied A twice per iteration: once
        # for the dual update, once more for the loss of the new iterate -- the next iteration's first residual)
        self.r = torch.empty_like(self.b)
        self.n_A = self.n_AT = 0          # calls of the user's operators (tests assert one of each per iteration)
        self._fid0 = torch.zeros((), dtype=torch.float64, device=self.device)
        self._residual(self.x, self._fid0)

    def _apply(self, op, v, shape, what):
        out = op(v)
        if tuple(out.shape) != tuple(shape) or out.dtype != self.dtype or not out.is_cuda:
            raise ValueError("%s must return a device tensor of shape %s and dtype %s, got %s %s on %s"
                             % (what, tuple(shape), self.dtype, tuple(out.shape), out.dtype, out.device))
        return out if out.is_contiguous() else out.contiguous()

    def _residual(self, x, fid_slot):
        """r <- A x - b, fid_slot <- 1/2 |r|^2 (this rank's share): one call of A, one HIP kernel."""
        ax = self._apply(self.A, x, self.b.shape, "A(x)")
        self.n_A += 1
        _nv.check(self.lib.tv_cpop_residual(_nv.dtype_code(self.dtype), self.b.numel(), _nv.ptr(ax), _nv.ptr(self.b), _nv.ptr(self.r),
                                            fid_slot.data_ptr(), _nv.ptr(self.ws), self.stream))

    def step(self, out):
        """out: fp64 device tensor [tv, fid] of this rank: tv = |D x|_{2,1} of the iterate the step started from, fid =
        1/2 |A x_new - b|^2 of the one it produced (the mix the reference's loss line uses, README.md:157).
        One call of A and one of A^T per iteration; every update of ours is a HIP kernel on preallocated buffers:
            tv_cpop_p        p <- (p + sigma_A r) / (1 + sigma_A)
            tv_cp_dual       q <- proj(q + sigma_D D x), TV
            tv_DT_axpy2      x_new <- x - tau A^T p - tau D^T q          (one pass: q, x and A^T p read once)
            tv_cpop_residual r <- A x_new - b, fidelity"""
        g, s = self.geo, self.slab
        h = self.plan.exchange_image(self.x, self.xh_prev, self.xh_next)      # in flight while the data-space update runs
        _nv.check(self.lib.tv_cpop_p(_nv.dtype_code(self.dtype), self.p.numel(), _nv.ptr(self.p), _nv.ptr(self.r), self.sigma_A, self.stream))
        if self.fused:
            # one sweep: A^T p first (the user's operator runs while the x halos travel), then dual + primal update in one pass
            # over q, the q' halos, the fix-up
            atp = self._apply(self.AT, self.p, self.x.shape, "AT(p)")
            self.n_AT += 1
            s.wait(h)
            _nv.check(self.lib.tv_cpop_fused(g.ref, _nv.ptr(self.x), _nv.ptr(self.xh_prev), _nv.ptr(self.xh_next), _nv.ptr(self.q),
                                             _nv.ptr(atp), _nv.ptr(self.x_new), self.sigma_D, self.reg, self.tau, 0, -1,
                                             out[0:1].data_ptr(), _nv.ptr(self.ws), self.stream))
            h = self.plan.exchange_grad(self.q, self.qh_prev[0] if self.qh_prev is not None else None,
                                        self.qh_next[0] if self.qh_next is not None else None)
            s.wait(h)
            _nv.check(self.lib.tv_cpop_fixup(g.ref, _nv.ptr(self.q), _nv.ptr(self.qh_prev), _nv.ptr(self.qh_next), _nv.ptr(self.x_new),
                                             self.tau, 0, -1, _nv.ptr(self.ws), self.stream))
            self.x, self.x_new = self.x_new, self.x
            self._residual(self.x, out[1:2])
            return
        s.wait(h)
        _nv.check(self.lib.tv_cp_dual(g.ref, _nv.ptr(self.x), _nv.ptr(self.xh_prev), _nv.ptr(self.xh_next), _nv.ptr(self.q),
                                      self.sigma_D, self.reg, out[0:1].data_ptr(), _nv.ptr(self.ws), self.stream))
        h = self.plan.exchange_grad(self.q, self.qh_prev[0] if self.qh_prev is not None else None,
                                    self.qh_next[0] if self.qh_next is not None else None)
        atp = self._apply(self.AT, self.p, self.x.shape, "AT(p)")             # the user's operator runs while the q halos travel
        self.n_AT += 1
        s.wait(h)
        _nv.check(self.lib.tv_DT_axpy2(g.ref, _nv.ptr(self.q), None, _nv.ptr(self.qh_prev), _nv.ptr(self.qh_next), _nv.ptr(self.x),
                                       _nv.ptr(atp), -self.tau, -self.tau, _nv.ptr(self.x_new), self.stream))
        self.x, self.x_new = self.x_new, self.x
        self._residual(self.x, out[1:2])

    def run(self, n_iter):
        hist = torch.zeros((n_iter, 2), dtype=torch.float64, device=self.device)
        for it in range(n_iter):
            self.step(hist[it])
        self.slab.allreduce_sum_(hist)
        h = hist.cpu().numpy()
        return h[:, 1] + self.reg * h[:, 0]

    def result(self):
        """The iterate as a DENSE (Nz, M, Ny, Nx) tensor: with padded state (``pitch``) a contiguous copy -- callers that ``.view(-1)``
        it, take its ``data_ptr()`` or hand it to the dense ``tv_*`` entry points must not see pad columns (round-4 advice); ``.x``
        stays the raw (possibly strided) view of the solver's storage."""
        return self.x.contiguous() if self.pitch != (0, 0) else self.x


# =================================================================================================
class SubgradientDescent(_SlabProblem):
    """README.md:118-124 with the state on the GPU: x <- x - step ((x - x0) + reg * G(x)).

    Per-step scalars: ``SLOTS`` fp64 device words, TV parts in [0:3], fidelity parts in [3:6] (one slot per launch:
    interior planes, first two planes, last two planes when the halo exchange is overlapped)."""

    SLOTS = 6

    @classmethod
    def loss_from_slots(cls, h, regularization):
        return h[:, 3:6].sum(axis=1) + regularization * h[:, 0:3].sum(axis=1)

    def __init__(self, x0, regularization, step_size, scheme="hybrid", reg_z_over_reg=1.0, reg_time=0.0,
                 mask_static=False, factor_reg_static=0, slab=None, one_pass=None, overlap=True, pitch="auto", tune_placement=None,
                 persistent=None):
        """tune_placement: None = on for one-pass problems (slabs) of >= 4 GiB per image with memory to spare (``_tune_placement``).
        persistent (round 6): None = volumes of at most ``SMALL_MAX_VOXELS`` on one GPU run ``run`` as ONE launch per ``SMALL_BLOCK``
        iterations (tv_small_subgrad_descent: tv_subgrad + tv_subgrad_step's arithmetic inside a persistent kernel); False / True."""
        super().__init__(x0, scheme, reg_z_over_reg, reg_time, mask_static, factor_reg_static, slab, pitch=pitch)
        self.reg, self.step_size = float(regularization), float(step_size)
        # (an explicit one_pass=True / False asks for that kernel family: the persistent loop replaces it only when it is asked for as well)
        self.small = self._small_ok(False if (one_pass is not None and persistent is None) else persistent)
        self.x = self.image_copy(self.x0)
        nz, m, ny, nx = self.x0.shape
        self.G = self.new_image()
        if one_pass is None:      # TV + G + step in one pass over x wherever the geometry allows (tv_subgrad_step_fused)
            one_pass = bool(self.lib.tv_subgrad_fused_supported(self.geo.ref))
            # the one-pass kernel folds the step into its store (it divides by step * reg): a vanishing product takes the two-pass path
            # (until round 5 the round-1 kernel took that corner for fp32; it lives in csrc/variants/ now)
            if self.step_size * self.reg < 1e-6:
                one_pass = False
        self.one_pass = bool(one_pass)
        # one pass: TV, G and the descent step in a single kernel, x ping-ponged (G is never stored); else the
        # two-pass tv_subgrad + tv_subgrad_step
        self.norms_ext = None if (self.one_pass and not self.small) else self.geo.new_image(nz + 2)
        if self.one_pass:
            self.x_alt = self.G
            self.G = None
        elif self.small:
            self.x_alt = self.new_image()           # the persistent loop ping-pongs the iterate
        self.ws = self.geo.workspace()
        self.plan = HaloPlan(self.slab, scheme, self.geo.z_active)
        sh = self.plan.on
        if sh and min(nz_r for _, nz_r in self.slab.parts) < 2:
            raise ValueError("the sub-gradient needs two halo planes: every rank must hold >= 2 planes")
        self.xh_prev = self.new_plane(2) if sh and self.slab.prev is not None else None
        self.xh_next = self.new_plane(2) if sh and self.slab.next is not None else None
        self.sh = sh
        # interior-first schedule (one-pass kernel: x is ping-ponged, so the planes being sent are never written while
        # the exchange is in flight): planes [2, nz - 2) need no halo and hide the two-plane exchange
        self.overlap = bool(overlap) and sh and self.one_pass and nz >= 5
        self._scratch = torch.zeros(self.SLOTS, dtype=torch.float64, device=self.device)
        self.placement = None
        if self.one_pass:
            img_bytes = self.x.numel() * self.x.element_size()
            if tune_placement is None:
                free, _total = torch.cuda.mem_get_info(self.device)
                tune_placement = img_bytes >= (4 << 30) and free >= 3 * img_bytes + (8 << 30)      # slabs too: local launches only
            if tune_placement:
                keep = (self.x, self.x_alt, self.x0)
                try:
                    self._tune_placement()
                except RuntimeError as exc:       # an optimisation, never a failure (out of memory while holding the candidates, ...)
                    self.x, self.x_alt, self.x0 = keep
                    self.x.copy_(self.x0)
                    torch.cuda.empty_cache()
                    self.placement = {"error": str(exc)[:200]}
                del keep

    def _tune_placement(self, n_extra=2, reps=2):
        """Pick where the two image buffers of the x ping-pong (and x0) live by MEASUREMENT, as ``ChambollePock._tune_x_placement``
        does and for the same reason (the time of a streaming kernel depends on where its arrays landed
        in physical memory, per allocation and per direction of the ping-pong; the descent loop on the north-star volume runs at
        128 - 149 it/s from process to process with the same binary).  Every ordered pair of 2 + ``n_extra`` candidates is timed with
        the real kernel; then the caller's x0 against a copy in a buffer that is left.  ~30 steps once per solver; the iterate is
        re-initialised afterwards, results do not depend on it."""
        import time as _time
        t_begin = _time.perf_counter()
        nz = self.slab.nz
        out = torch.zeros(self.SLOTS, dtype=torch.float64, device=self.device)
        hp = self.xh_prev if self.xh_prev is not None else None
        hn = self.xh_next if self.xh_next is not None else None

        def one(i_buf, o_buf):
            self.x, self.x_alt = i_buf, o_buf
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self._one_pass_range(0, nz, hp, hn, out[0:1], out[3:4])
            b.record()
            return a, b

        def round_trip(u, v):
            evs = [(one(u, v), one(v, u)) for _ in range(reps)]
            torch.cuda.synchronize(self.device)
            return min(e[0][0].elapsed_time(e[0][1]) for e in evs) + min(e[1][0].elapsed_time(e[1][1]) for e in evs)

        x_a, x_b = self.x, self.x_alt
        for _ in range(2):                                                   # warm-up (code load, clocks)
            one(x_a, x_b)
        cands = [x_a, x_b] + [self.new_image() for _ in range(n_extra)]
        for c in cands[1:]:
            c.copy_(self.x0)                                                 # every candidate holds data of the problem's kind (no denormals, no NaN)
        n = len(cands)
        ev = [[[] for _ in range(n)] for _ in range(n)]
        for r in range(reps):
            for i in range(n):
                for j in range(n):
                    if i != j:
                        ev[i][j].append(one(cands[i], cands[j]))
        torch.cuda.synchronize(self.device)
        t = [[(min(a.elapsed_time(b) for a, b in ev[i][j]) if i != j else float("inf")) for j in range(n)] for i in range(n)]
        _, bi, bj = min(((t[i][j] + t[j][i], i, j) for i in range(n) for j in range(i + 1, n)))
        info = {"candidates": n, "step_ms": [[None if i == j else round(t[i][j], 3) for j in range(n)] for i in range(n)],
                "chosen": [bi, bj], "chosen_ms": [round(t[bi][bj], 3), round(t[bj][bi], 3)], "first_pair_ms": [round(t[0][1], 3), round(t[1][0], 3)]}
        x_a, x_b = cands[bi], cands[bj]
        spare = [c for k, c in enumerate(cands) if k not in (bi, bj)]
        del cands
        if spare:                                                            # x0 (read once per step): the caller's array against a copy
            x0_orig, x0_copy = self.x0, spare[0]
            x0_copy.copy_(x0_orig)
            t0_ = round_trip(x_a, x_b)
            self.x0 = x0_copy
            t1_ = round_trip(x_a, x_b)
            info["x0_round_trip_ms"] = [round(t0_, 3), round(t1_, 3)]
            if t0_ <= t1_:
                self.x0 = x0_orig
            del x0_orig, x0_copy
        del spare
        self.x, self.x_alt = x_a, x_b
        self.x.copy_(self.x0)
        torch.cuda.synchronize(self.device)
        torch.cuda.empty_cache()
        info["seconds"] = round(_time.perf_counter() - t_begin, 3)
        self.placement = info

    def _one_pass_range(self, a, b, hp, hn, tv_slot, fid_slot):
        g, x = self.geom(a, b), self.x
        _nv.check(self.lib.tv_subgrad_step_fused(g.ref, _nv.ptr(x[a:b]), _nv.ptr(hp), _nv.ptr(hn), _nv.ptr(self.x0[a:b]),
                                                 _nv.ptr(self.x_alt[a:b]), self.step_size, self.reg, tv_slot.data_ptr(),
                                                 fid_slot.data_ptr(), _nv.ptr(self.ws), self.stream))

    def step(self, out=None):
        """Enqueue one iteration; out: fp64 device tensor of SLOTS words (TV parts [0:3], fidelity parts [3:6])."""
        out = self._scratch if out is None else out
        nz, s, x = self.slab.nz, self.slab, self.x
        h = self.plan.exchange_image2(x, self.xh_prev, self.xh_next)
        if self.one_pass:
            if self.overlap:
                self._one_pass_range(2, nz - 2, x[0:2], x[nz - 2:nz], out[0:1], out[3:4])
                s.wait(h)
                self._one_pass_range(0, 2, self.xh_prev, x[2:4], out[1:2], out[4:5])
                self._one_pass_range(nz - 2, nz, x[nz - 4:nz - 2], self.xh_next, out[2:3], out[5:6])
            else:
                s.wait(h)
                self._one_pass_range(0, nz, self.xh_prev, self.xh_next, out[0:1], out[3:4])
            self.x, self.x_alt = self.x_alt, self.x
            return
        s.wait(h)
        g = self.geo
        _nv.check(self.lib.tv_subgrad(g.ref, _nv.ptr(x), _nv.ptr(self.xh_prev), _nv.ptr(self.xh_next), _nv.ptr(self.G),
                                      _nv.ptr(self.norms_ext), out[0:1].data_ptr(), _nv.ptr(self.ws), self.stream))
        _nv.check(self.lib.tv_subgrad_step(g.ref, _nv.ptr(x), _nv.ptr(self.x0), _nv.ptr(self.G), self.step_size, self.reg,
                                           out[3:4].data_ptr(), _nv.ptr(self.ws), self.stream))

    GRAPH_BLOCK = 10            # iterations captured per hipGraph (even: the x ping-pong returns to its start)
    GRAPH_MAX_VOXELS = 1 << 23  # below this an iteration is launch-bound (a few tens of microseconds of kernels)

    def run(self, n_iter, graph=None):
        """n_iter iterations; returns the README's loss history (README.md:124).  graph: None = replay blocks of
        GRAPH_BLOCK iterations from a hipGraph when the problem is small enough to be launch-bound and not sharded
        (the README's own 2-D example is), True / False force it."""
        hist = torch.zeros((n_iter, self.SLOTS), dtype=torch.float64, device=self.device)
        if self.small and graph is None:
            self._run_small(hist)
            return self._small_check(self.loss_from_slots(hist.cpu().numpy(), self.reg))
        use_graph = (self.x0.numel() <= self.GRAPH_MAX_VOXELS) if graph is None else bool(graph)
        start = 0
        if use_graph and not self.slab.sharded and n_iter >= 2 + 2 * self.GRAPH_BLOCK:
            self.step(hist[0])
            self.step(hist[1])
            start = 2 + self._run_graphed_from(hist, 2, n_iter)
        for it in range(start, n_iter):
            self.step(hist[it])
        self.slab.allreduce_sum_(hist)
        return self.loss_from_slots(hist.cpu().numpy(), self.reg)

    def _run_small(self, rows):
        """``len(rows)`` descent steps in blocks of ``SMALL_BLOCK`` per cooperative launch (tv_small_subgrad_descent); TV(x_k) -> slot 0,
        1/2 |x_{k+1} - x0|^2 -> slot 3.  The iterate is ping-ponged: after an odd block x and x_alt trade places."""
        ws = self._small_ws()
        n, done = rows.shape[0], 0
        while done < n:
            k = min(self.SMALL_BLOCK, n - done)
            _nv.check(self.lib.tv_small_subgrad_descent(self.geo.ref, _nv.ptr(self.x), _nv.ptr(self.x_alt), _nv.ptr(self.x0), _nv.ptr(self.norms_ext),
                                                        self.step_size, self.reg, k, rows[done].data_ptr(), rows.stride(0), 3, _nv.ptr(ws), self.stream))
            if k & 1:
                self.x, self.x_alt = self.x_alt, self.x
            done += k

    def _run_graphed_from(self, hist, first, n_iter):
        """Capture GRAPH_BLOCK iterations (not executed during capture) and replay them over hist[first:]."""
        K = self.GRAPH_BLOCK
        nrep = (n_iter - first) // K
        if nrep < 1:
            return 0
        x_ref, xalt_ref = self.x, getattr(self, "x_alt", None)
        try:
            buf = torch.zeros((K, self.SLOTS), dtype=torch.float64, device=self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for k in range(K):
                    self.step(buf[k])
        except Exception:
            self.x = x_ref                                 # nothing ran: undo the bookkeeping, stay eager
            if xalt_ref is not None:
                self.x_alt = xalt_ref
            return 0
        done = 0
        for r in range(nrep):
            graph.replay()
            hist[first + done:first + done + K].copy_(buf)
            done += K
        return done

    def result(self):
        """The iterate as a DENSE (Nz, M, Ny, Nx) tensor: with padded state (``pitch``) a contiguous copy -- callers that ``.view(-1)``
        it, take its ``data_ptr()`` or hand it to the dense ``tv_*`` entry points must not see pad columns (round-4 advice); ``.x``
        stays the raw (possibly strided) view of the solver's storage."""
        return self.x.contiguous() if self.pitch != (0, 0) else self.x


# =================================================================================================
class ADMM(_SlabProblem):
    """Scaled-form ADMM for min 1/2|x-x0|^2 + reg |z|_{2,1} s.t. Dx = z  (the
    reference names ADMM, README.md:26,135, but ships no code -- parity is pinned against
    oracle/tv_oracle.py::admm and, op by op, against the reference's D / D^T).

      x-step : (I + rho D^T D) x = x0 + rho D^T (z - u)   n_cg CG steps, warm start
      z-step : z = shrink(Dx + u, reg/rho);  u-step: u += Dx - z          (one fused kernel)

    single_reduction (default): the CG recurrence in its Chronopoulos-Gear form -- w = A r, gamma = <r,r> and
    delta = <r,w> come out of ONE kernel and are all-reduced together (one collective of two scalars per CG step instead
    of two collectives), the four vector updates are one kernel, the residual of the warm start and the loss of the
    outer iteration are fused into the kernels that read those vectors anyway, and the dual pair is kept as
    (t = z - u, u) so that the right-hand side reads one gradient array: 4 Nd + 11 n_cg + 5 words per voxel and outer
    iteration instead of 5 Nd + 11 n_cg + 17 (76 instead of 92 for Nd = 4, n_cg = 5).  ``single_reduction=False`` is the
    textbook recurrence of round 1 (tv_cg_step1 / tv_cg_step2).  Both are the same iteration in exact arithmetic;
    oracle.admm restates both.

    DEFAULTS THAT CHANGED IN ROUND 4 (existing callers get different -- equally valid -- iterates): ``x_solver=None`` runs the x-solve
    as ``n_cg`` Chebyshev steps instead of CG (same objective to 6 - 7 digits at every outer iteration, profiles/
    r4_admm_xsolve_study_f32.txt; ``x_solver="cg"`` is the old behaviour), ``keep_z=True`` stores every sample of t' so that ``.z`` works on
    the one-sweep path (``keep_z=False`` moves Nd fewer words per voxel), and ``pitch="auto"`` pads ragged rows: ``.x`` is then a strided
    view of padded storage, ``result()`` a dense copy."""

    def __init__(self, x0, regularization, rho, n_cg=10, scheme="hybrid", reg_z_over_reg=1.0, reg_time=0.0,
                 mask_static=False, factor_reg_static=0, slab=None, single_reduction=True, fused=None, keep_z=True, x_solver=None,
                 pitch="auto", tune_placement=None):
        """tune_placement: None = on for unsharded problems whose state is >= 16 GiB with room for a second copy (``_tune_placement``)."""
        super().__init__(x0, scheme, reg_z_over_reg, reg_time, mask_static, factor_reg_static, slab, pitch=pitch)
        self.reg, self.rho, self.n_cg = float(regularization), float(rho), int(n_cg)
        self.single = bool(single_reduction)
        # x_solver="chebyshev" (round 3): n_cg steps of the Chebyshev iteration on (I + rho D^T D) e = b - A x instead of CG.  Its
        # scalars follow from the spectral interval [1, 1 + rho L] alone, so a step is ONE streaming kernel (tv_cheb_step: e_k with
        # its stencil, r0 and e_{k-1} read, e_{k+1} written -- 4 words per voxel where a CG step moves 11) and a sharded solve
        # exchanges halo planes only: no all-reduce.  Same convergence as CG on this operator within the digits the loss is
        # reported to (the spectrum of D^T D fills its interval); oracle.admm(x_solver="chebyshev") restates it.
        # DEFAULT since round 4 (x_solver=None): Chebyshev wherever it applies (single_reduction, n_cg > 0), CG otherwise.  Measured on
        # the configs[4] per-GPU slab, 50 outer iterations, four schemes, rho in {0.02, 0.05, 0.2}, fp32 and fp64
        # (tools/archive/admm_xsolve_study.py, profiles/r4_admm_xsolve_study_*.txt): with the same number of steps the two reach the same
        # objective to 6 - 7 digits at EVERY outer iteration, and a Chebyshev outer iteration takes 12.8 - 16.3 ms where CG takes
        # 27 - 33 (fp64: 15 - 21 against 29 - 37): 2.1 - 2.4 x less wall time to any objective level, no all-reduce in the solve.
        # ``n_cg`` keeps its meaning: steps of the x-solve per outer iteration.  x_solver="cg" is the round-1..3 behaviour.
        if x_solver is None:
            x_solver = "chebyshev" if (self.single and self.n_cg > 0) else "cg"
        if x_solver not in ("cg", "chebyshev"):
            raise ValueError("x_solver must be None (the default since round 4: 'chebyshev' wherever it applies, i.e. single_reduction and "
                             "n_cg > 0), 'cg' (the behaviour of rounds 1 - 3) or 'chebyshev'")
        self.cheb = (x_solver == "chebyshev")
        if self.cheb and not (self.single and self.n_cg > 0):
            raise ValueError("x_solver='chebyshev' needs single_reduction=True (the default) and n_cg > 0")
        L = normal_spectral_bound(scheme, self.slab.nz_global, x0.shape[1], reg_z_over_reg, reg_time, self.geo.time_weight_max)
        self._cheb_coef = chebyshev_coefficients(1.0 + self.rho * L, self.n_cg) if self.cheb else None
        # one-sweep dual side (round 3): z / u update + the residual of the next x-solve in one pass over u
        # (tv_admm_sweep + tv_admm_fixup: 2 Nd + 3 words per voxel instead of the 4 Nd + 6 of tv_admm_tu + tv_DT_axpy +
        # tv_normal_op2).  keep_z=True (default since round 4: ``.z`` keeps working as it did before the one-sweep path existed): round 5
        # keeps a SECOND u array instead of every sample of t' -- u is ping-ponged and z = shrink(D x + u_old) is rebuilt when ``.z`` is
        # asked for, so the sweep moves the same 2 Nd + 3 words as with keep_z=False (round 4: 3 Nd + 3); keep_z=False saves the memory
        # of that array -- the split variable z is then not recoverable and ``.z`` raises, naming this flag.
        can_fuse = self.single and self.n_cg > 0 and bool(self.lib.tv_cp_fused_supported(self.geo.ref))
        if fused and not can_fuse:
            raise ValueError("fused=True needs single_reduction, n_cg > 0 and a geometry tv_cp_fused_supported() accepts")
        self.fused = can_fuse if fused is None else bool(fused)
        self.keep_z = bool(keep_z)
        self._have_r = False
        self.x = self.image_copy(self.x0)
        self._zt = self.new_grad()                   # z, or t = z - u (single_reduction)
        self.u = self.new_grad()
        # round 5: on the one-sweep path keep_z costs MEMORY, not traffic -- u is ping-ponged (tv_admm_sweep reads u, writes u_alt, the roles
        # swap), so z = shrink(D x + u_old) can be rebuilt whenever ``.z`` is asked for and the sweep stores t' sparsely as with
        # keep_z=False: 2 Nd + 3 words per voxel instead of 3 Nd + 3 (round 4 stored every sample of t' = z - u - D x)
        self.u_alt = self.new_grad() if (self.fused and self.keep_z) else None
        self.b = self.new_image()
        self.r = self.new_image()
        self.d = self.new_image()
        self.Ad = self.new_image()                   # A d (textbook) / s = A d (single reduction)
        self.ws = self.geo.workspace()
        self.plan = HaloPlan(self.slab, scheme, self.geo.z_active)
        pl, s_ = self.plan, self.slab
        sh = pl.on
        if sh and min(nz_r for _, nz_r in s_.parts) < 2:
            raise ValueError("the normal operator needs two halo planes: every rank must hold >= 2 planes")
        self.sh = sh
        self.h2_prev = self.new_plane(2) if sh and s_.prev is not None else None
        self.h2_next = self.new_plane(2) if sh and s_.next is not None else None
        self.ch_back, self.ch_fwd = pl.ch_back, pl.ch_fwd
        self.wh_prev = self.new_plane() if pl.g_need_prev else None
        self.wh_next = self.new_plane() if pl.g_need_next else None
        self.ws_prev = self.new_plane() if pl.g_send_prev else None
        self.ws_next = self.new_plane() if pl.g_send_next else None
        self.sc = torch.zeros(4, dtype=torch.float64, device=self.device)   # rs, dAd, rs_new, spare / gamma, delta, gamma_old, alpha_old
        self.dots3 = torch.zeros((3, 2), dtype=torch.float64, device=self.device)
        self.dots = torch.zeros(2, dtype=torch.float64, device=self.device)
        self.rr = torch.zeros(2, dtype=torch.float64, device=self.device)     # <r, r> of the sweep / of the fix-up
        self.timing = None      # set to a list to collect one list of (name, HIP event) per outer iteration (bench.py --solver admm)
        self._marks = None
        self.placement = None
        # (u_alt is a candidate of the tuner too: count it, or `free >= set_bytes + 8 GiB` under-estimates by Nd words -- round-5 advice)
        set_bytes = (sum(getattr(self, k).numel() for k in self._STATE) + (self.u_alt.numel() if self.u_alt is not None else 0)) * self.x.element_size()
        if tune_placement is None:
            free, _total = torch.cuda.mem_get_info(self.device) if self.device.type == "cuda" else (0, 0)
            tune_placement = (not self.slab.sharded) and set_bytes >= (16 << 30) and free >= set_bytes + (8 << 30)
        if tune_placement:
            self._tune_placement()

    _STATE = ("x", "_zt", "u", "b", "r", "d", "Ad")      # the arrays an outer iteration streams through

    # ---- what the x-solve of one outer iteration moves (bench.py's roofline_xsolve) ----------------------------------------------
    @property
    def xsolve_words(self):
        """Algorithmic words per voxel of the x-solve of ONE outer iteration (every array a launch touches counted once).
        Chebyshev, K = n_cg steps: e_1 = a0 r is never stored -- e_2 from r alone (2), a step with y = a0 r formed on the fly (3),
        K - 4 plain steps (e_k with its stencil, r, e_{k-1} read; e_{k+1} written: 4) and a last step that adds x (5): 4 K - 6.
        CG (single reduction): normal operator on r (2) + the fused update of four vectors (9) per step."""
        K = self.n_cg
        if self.cheb:
            return 3 if K == 1 else 4 * K - 6
        return 11 * K if self.single else 11 * K + 4

    @property
    def xsolve_launches(self):
        K = self.n_cg
        if self.cheb:
            return 1 if K == 1 else K - 1
        return 2 * K

    @property
    def xsolve_desc(self):
        if self.cheb:
            return ("tv_cheb_step: k_normal_stream<M,TWIN,T,CHEB> (%d launches: e_{k+1} = e_k + alpha (r - (I + rho D^T D) e_k) + beta (e_k - e_{k-1}), "
                    "one streaming pass each)" % self.xsolve_launches)
        return "tv_normal_op2 + tv_cg_update (%d launches)" % self.xsolve_launches

    def _reset_state(self):
        """Back to the state of a fresh solver (x = x0, everything else zero) in the arrays that are bound now."""
        self.x.copy_(self.x0)
        for k in self._STATE[1:]:
            getattr(self, k).zero_()
        if self.u_alt is not None:
            self.u_alt.zero_()
        for t in (self.sc, self.dots3, self.dots, self.rr):
            t.zero_()
        self._have_r = False

    def _tune_placement(self, n_sets=4, n_steps=2):
        """Pick WHERE the state lives by measurement, as ``ChambollePock._tune_x_placement`` does: the same
        outer iteration on the same data takes 17.5 or 19.5 ms on the configs[4] slab depending on where u, t and the image buffers
        landed (tools/archive/admm_placement_probe.py).  ``n_sets`` complete sets of state arrays are allocated one after the other (never
        more than two alive), ``n_steps`` outer iterations are timed on each after one to warm up, the fastest set is kept and put
        back into the initial state.  Unsharded problems only (an outer iteration of a slab waits for its neighbours)."""
        import time as _time
        t_begin = _time.perf_counter()
        out = torch.zeros((n_steps + 1, 2), dtype=torch.float64, device=self.device)

        def timed():
            self._reset_state()
            self.step(out[0])
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for k in range(n_steps):
                self.step(out[1 + k])
            b.record()
            torch.cuda.synchronize(self.device)
            return a.elapsed_time(b) / n_steps

        names = self._STATE + (("u_alt",) if self.u_alt is not None else ())

        def bound():                                 # the buffer roles rotate inside a step: take what is bound NOW
            return {k: getattr(self, k) for k in names}

        # the WHOLE tuner is an optimisation, never a failure (round-4 advice: the baseline measurement used to sit outside the try --
        # an out-of-memory error in a halo or Chebyshev temporary there failed the constructor)
        best, best_t, times, info = bound(), float("inf"), [], {}
        try:
            best_t = timed()
            best = bound()
            times.append(round(best_t, 3))
            for _ in range(n_sets - 1):
                cand = dict(x=self.new_image(), _zt=self.new_grad(), u=self.new_grad(), b=self.new_image(), r=self.new_image(),
                            d=self.new_image(), Ad=self.new_image())
                if self.u_alt is not None:
                    cand["u_alt"] = self.new_grad()
                for k, v in cand.items():
                    setattr(self, k, v)
                t = timed()
                times.append(round(t, 3))
                cand = bound()
                if t < best_t:
                    best_t, best = t, cand
                del cand
                for k, v in best.items():            # the loser loses its last reference here
                    setattr(self, k, v)
                torch.cuda.empty_cache()
        except RuntimeError as exc:                  # out of memory while a second set was alive, ...: an optimisation, never a failure
            for k, v in best.items():
                setattr(self, k, v)
            torch.cuda.empty_cache()
            info["error"] = str(exc)[:200]
        self._reset_state()
        torch.cuda.synchronize(self.device)
        info.update({"outer_ms": times, "chosen": times.index(min(times)) if times else None, "seconds": round(_time.perf_counter() - t_begin, 3)})
        self.placement = info

    @property
    def z(self):
        """The split variable z (single_reduction keeps t = z - u: z = t + u; the one-sweep path keeps t' = t - D x and recomputes
        D x -- on a sharded slab that is a halo exchange, so every rank has to ask for z, not just one).
        MEMORY: on the one-sweep path each access allocates two transient gradient-sized arrays (2 Nd words per voxel: z itself and a
        scratch copy of the dual variable) -- 32 GiB at the config4 hybrid slab; they are released when the caller drops z."""
        if self.fused and self._have_r:
            if not self.keep_z:
                raise RuntimeError("ADMM(..., keep_z=False) stores only the samples of z - u - D x its fix-up reads: z is not available; "
                                   "construct the solver with keep_z=True (the default)")
            # rebuilt on demand from x and the dual variable the last sweep READ (u_alt after the swap): z = shrink(D x + u_old)
            z = self.new_grad()
            u_old = self.new_grad()                  # (not .clone(): a clone of a pitched view is dense)
            u_old.copy_(self.u_alt)
            tv_ = torch.zeros(1, dtype=torch.float64, device=self.device)
            hp, hn = self._halo2(self.x)
            _nv.check(self.lib.tv_admm_zu(self.geo.ref, _nv.ptr(self.x), _nv.ptr(hp[1:2] if hp is not None else None),
                                          _nv.ptr(hn[0:1] if hn is not None else None), _nv.ptr(z), _nv.ptr(u_old), self.reg / self.rho,
                                          tv_.data_ptr(), _nv.ptr(self.ws), self.stream))
            return z
        return self._zt + self.u if self.single else self._zt

    def _halo2(self, v):
        self.slab.wait(self.plan.exchange_image2(v, self.h2_prev, self.h2_next))
        return self.h2_prev, self.h2_next

    def _normal_range(self, v, out, a, b, hp, hn, dots, rhs=None, out2=None):
        """out[a:b] = (I + rho D^T D) v (or rhs - that, with a copy in out2) on local planes [a, b); hp / hn: two-plane
        halos of that range; dots: two device words."""
        g = self.geom(a, b)
        _nv.check(self.lib.tv_normal_op2(g.ref, _nv.ptr(v[a:b]), _nv.ptr(hp), _nv.ptr(hn), self.rho,
                                         _nv.ptr(rhs[a:b]) if rhs is not None else None, _nv.ptr(out[a:b]),
                                         _nv.ptr(out2[a:b]) if out2 is not None else None, dots.data_ptr(), _nv.ptr(self.ws),
                                         self.stream))

    def _normal(self, v, out, dots, rhs=None, out2=None, reduce=True):
        """Two-plane halo exchange hidden behind the interior planes, then the operator; dots (2 device words) summed
        over the launches and, with ``reduce``, over the ranks."""
        nz = self.slab.nz
        if self.sh and nz >= 5:
            h = self.plan.exchange_image2(v, self.h2_prev, self.h2_next)
            self._normal_range(v, out, 2, nz - 2, v[0:2], v[nz - 2:nz], self.dots3[0], rhs, out2)
            self.slab.wait(h)
            self._normal_range(v, out, 0, 2, self.h2_prev, v[2:4], self.dots3[1], rhs, out2)
            self._normal_range(v, out, nz - 2, nz, v[nz - 4:nz - 2], self.h2_next, self.dots3[2], rhs, out2)
            torch.sum(self.dots3, dim=0, out=dots)
        else:
            hp, hn = self._halo2(v)
            self._normal_range(v, out, 0, nz, hp, hn, dots, rhs, out2)
        if reduce:
            self.slab.allreduce_sum_(dots)

    def _cheb_range(self, v, y, yscale, add, ref, out, alpha, beta, a, b, hp, hn, dots):
        g = self.geom(a, b)
        sl = lambda t_: _nv.ptr(t_[a:b]) if t_ is not None else None      # noqa: E731
        _nv.check(self.lib.tv_cheb_step(g.ref, _nv.ptr(v[a:b]), _nv.ptr(hp), _nv.ptr(hn), self.rho, _nv.ptr(self.r[a:b]), sl(y), yscale,
                                        sl(add), sl(ref), alpha, beta, _nv.ptr(out[a:b]), dots.data_ptr() if dots is not None else None,
                                        _nv.ptr(self.ws), self.stream))

    def _cheb_step(self, v, y, yscale, add, ref, out, alpha, beta, dots):
        """out = [add +] v + alpha (r - A v) + beta (v - y) on the slab (y None: y = yscale r); the two-plane halo exchange of v hides
        behind the interior planes exactly as in _normal; dots (or None: not wanted) <- [|r - A v|^2, |out - ref|^2 or |v|^2] summed over the
        launches."""
        nz = self.slab.nz
        args = (y, yscale, add, ref, out, alpha, beta)
        if self.sh and nz >= 5:
            h = self.plan.exchange_image2(v, self.h2_prev, self.h2_next)
            d3 = self.dots3 if dots is not None else (None, None, None)
            self._cheb_range(v, *args, 2, nz - 2, v[0:2], v[nz - 2:nz], d3[0])
            self.slab.wait(h)
            self._cheb_range(v, *args, 0, 2, self.h2_prev, v[2:4], d3[1])
            self._cheb_range(v, *args, nz - 2, nz, v[nz - 4:nz - 2], self.h2_next, d3[2])
            if dots is not None:
                torch.sum(self.dots3, dim=0, out=dots)
        else:
            hp, hn = self._halo2(v)
            self._cheb_range(v, *args, 0, nz, hp, hn, dots)

    def _solve_cheb(self, fid_slot):
        """x <- x + e_K, e_K = K Chebyshev steps on A e = r (self.r = b - A x); fid_slot <- |x - x0|^2 (local), or None if the caller
        takes the fidelity from the sweep that follows."""
        g, lib, K, coef = self.geo, self.lib, self.n_cg, self._cheb_coef
        bufs = [self.d, self.Ad, self.b]                  # all free between two solves
        a0 = coef[0][0]
        ref = self.x0 if fid_slot is not None else None
        if K == 1:
            _nv.check(lib.tv_axpby(g.ref, a0, _nv.ptr(self.r), 1.0, _nv.ptr(self.x), _nv.ptr(ref), _nv.ptr(bufs[0]),
                                   fid_slot.data_ptr() if fid_slot is not None else None, _nv.ptr(self.ws) if fid_slot is not None else None,
                                   self.stream))
            last = 0
        else:
            # e_1 = a0 r is never stored: the first kernel forms e_2 = e_1 + alpha_1 (r - A e_1) + beta_1 e_1 from r alone (x = b = r:
            # e_2 = r + a' (r - A r) + b' r), the second one takes y = a0 r on the fly -- 2 + 3 words instead of 2 + 3 + 4
            al1, be1 = coef[1]
            ap = al1 * a0
            bp = a0 + al1 + be1 * a0 - 1.0 - ap
            fin = (K == 2)
            # the dot products of a step (|r - A e_k|^2, |out - x0|^2) are asked for where they are used: by the last step, for the fidelity
            want = lambda fin_: self.dots if (fin_ and ref is not None) else None      # noqa: E731
            self._cheb_step(self.r, None, 0.0, self.x if fin else None, ref if fin else None, bufs[0], ap, bp, want(fin))
            for k in range(2, K):         # e_k lives in bufs[(k - 2) % 3]
                fin = (k + 1 == K)
                alpha, beta = coef[k]
                self._cheb_step(bufs[(k - 2) % 3], bufs[(k - 3) % 3] if k >= 3 else None, a0 if k == 2 else 0.0,
                                self.x if fin else None, ref if fin else None, bufs[(k - 1) % 3], alpha, beta, want(fin))
            if fid_slot is not None:
                fid_slot.copy_(self.dots[1:2])
            last = (K - 2) % 3
        # the new image was written next to the old one: swap the roles (the old x becomes a scratch vector)
        self.x, bufs[last] = bufs[last], self.x
        self.d, self.Ad, self.b = bufs

    def _rhs(self):
        """b = x0 + rho D^T (z - u); the boundary planes of (z - u) travel to the neighbours first."""
        g, lib, s, nz = self.geo, self.lib, self.slab, self.slab.nz
        code, plane = _nv.dtype_code(self.dtype), g.plane
        if self.single:          # the first array IS t = z - u
            h = s.exchange(send_prev=self._zt[0, self.ch_fwd] if self.plan.g_send_prev else None,
                           send_next=self._zt[nz - 1, self.ch_back] if self.plan.g_send_next else None,
                           recv_prev=self.wh_prev[0] if self.wh_prev is not None else None,
                           recv_next=self.wh_next[0] if self.wh_next is not None else None)
            s.wait(h)
            _nv.check(lib.tv_DT_axpy(g.ref, _nv.ptr(self._zt), None, _nv.ptr(self.wh_prev), _nv.ptr(self.wh_next),
                                     _nv.ptr(self.x0), self.rho, _nv.ptr(self.b), self.stream))
            return
        if self.plan.g_send_next:
            _nv.check(lib.tv_sub(code, plane, _nv.ptr(self._zt[nz - 1, self.ch_back]), _nv.ptr(self.u[nz - 1, self.ch_back]),
                                 _nv.ptr(self.ws_next), self.stream))
        if self.plan.g_send_prev:
            _nv.check(lib.tv_sub(code, plane, _nv.ptr(self._zt[0, self.ch_fwd]), _nv.ptr(self.u[0, self.ch_fwd]),
                                 _nv.ptr(self.ws_prev), self.stream))
        s.wait(s.exchange(send_prev=self.ws_prev, send_next=self.ws_next, recv_prev=self.wh_prev, recv_next=self.wh_next))
        _nv.check(lib.tv_DT_axpy(g.ref, _nv.ptr(self._zt), _nv.ptr(self.u), _nv.ptr(self.wh_prev), _nv.ptr(self.wh_next),
                                 _nv.ptr(self.x0), self.rho, _nv.ptr(self.b), self.stream))

    def _zu(self, out_tv):
        """z / u update (needs one x halo plane per side: the two-plane exchange is reused)."""
        g, lib = self.geo, self.lib
        hp, hn = self._halo2(self.x)
        xp = hp[1:2] if hp is not None else None      # plane z0-1
        xn = hn[0:1] if hn is not None else None      # plane z0+nz
        fn = lib.tv_admm_tu if self.single else lib.tv_admm_zu
        _nv.check(fn(g.ref, _nv.ptr(self.x), _nv.ptr(xp), _nv.ptr(xn), _nv.ptr(self._zt), _nv.ptr(self.u),
                     self.reg / self.rho, out_tv.data_ptr(), _nv.ptr(self.ws), self.stream))

    def step(self, out):
        """One outer iteration; out: fp64 device tensor [tv, |x - x0|^2] of this rank."""
        if self.single:
            return self._step_single(out)
        g, lib, s = self.geo, self.lib, self.slab
        code = _nv.dtype_code(self.dtype)
        self._rhs()
        # ---- CG on (I + rho D^T D) x = b, textbook recurrence -----------------------------------------
        rs, dAd, rs_new = self.sc[0:1], self.sc[1:2], self.sc[2:3]
        self._normal(self.x, self.r, self.dots, rhs=self.b, out2=self.d)      # r = d = b - A x, dots[0] = <r, r>
        rs.copy_(self.dots[0:1])
        for _ in range(self.n_cg):
            self._normal(self.d, self.Ad, self.dots)
            dAd.copy_(self.dots[0:1])
            _nv.check(lib.tv_cg_step1(g.ref, _nv.ptr(self.x), _nv.ptr(self.r), _nv.ptr(self.d), _nv.ptr(self.Ad),
                                      rs.data_ptr(), dAd.data_ptr(), rs_new.data_ptr(), _nv.ptr(self.ws), self.stream))
            s.allreduce_sum_(rs_new)
            _nv.check(lib.tv_cg_step2(g.ref, _nv.ptr(self.d), _nv.ptr(self.r), rs_new.data_ptr(), rs.data_ptr(), self.stream))
            rs.copy_(rs_new)
        self._zu(out[0:1])
        # fidelity 1/2 |x - x0|^2 = 1/2 <x-x0, x-x0>: r is free now
        _nv.check(lib.tv_sub(code, g.image_elems, _nv.ptr(self.x), _nv.ptr(self.x0), _nv.ptr(self.r), self.stream))
        _nv.check(lib.tv_dot(g.ref, _nv.ptr(self.r), _nv.ptr(self.r), out[1:2].data_ptr(), _nv.ptr(self.ws), self.stream))

    def _zu_fused(self, out_tv, out_fid=None):
        """z / u update and r = b - A x of the next solve in one sweep + fix-up; gamma = <r, r> (local) into sc[0] -- or, with the
        Chebyshev x-solve (which needs no gamma), |x - x0|^2 of the iterate into out_fid."""
        g, lib, s, nz = self.geo, self.lib, self.slab, self.slab.nz
        hp, hn = self._halo2(self.x)
        xp = hp[1:2] if hp is not None else None      # plane z0-1
        xn = hn[0:1] if hn is not None else None      # plane z0+nz
        u_out = self.u_alt if self.u_alt is not None else self.u          # keep_z: ping-pong (z is rebuilt from the array that was READ)
        _nv.check(lib.tv_admm_sweep(g.ref, _nv.ptr(self.x), _nv.ptr(xp), _nv.ptr(xn), _nv.ptr(self.u), _nv.ptr(u_out), _nv.ptr(self._zt),
                                    _nv.ptr(self.x0), _nv.ptr(self.r), self.reg / self.rho, self.rho,
                                    (2 if out_fid is not None else 0),
                                    0, -1, out_tv.data_ptr(), (out_fid if out_fid is not None else self.rr[0:1]).data_ptr(),
                                    _nv.ptr(self.ws), self.stream))
        if self.u_alt is not None:
            self.u, self.u_alt = self.u_alt, self.u
        self._mark("sweep")
        # the boundary planes of t' travel to the neighbours (as those of t = z - u do in _rhs)
        h = s.exchange(send_prev=self._zt[0, self.ch_fwd] if self.plan.g_send_prev else None,
                       send_next=self._zt[nz - 1, self.ch_back] if self.plan.g_send_next else None,
                       recv_prev=self.wh_prev[0] if self.wh_prev is not None else None,
                       recv_next=self.wh_next[0] if self.wh_next is not None else None)
        s.wait(h)
        _nv.check(lib.tv_admm_fixup(g.ref, _nv.ptr(self._zt), _nv.ptr(self.wh_prev), _nv.ptr(self.wh_next), _nv.ptr(self.r), self.rho,
                                    0, -1, self.rr[1:2].data_ptr(), _nv.ptr(self.ws), self.stream))
        if out_fid is None:
            torch.sum(self.rr, dim=0, keepdim=True, out=self.sc[0:1])
        self._have_r = True

    def _mark(self, name):
        """A HIP event on the launch stream (torch's current stream == the stream the C-ABI enqueues on), only while ``timing`` is a list."""
        if self._marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._marks.append((name, e))

    def _step_single(self, out):
        if self.timing is None:
            return self._step_single_body(out)
        self._marks = []
        self._mark("start")
        self._step_single_body(out)
        self._mark("end")
        self.timing.append(self._marks)
        self._marks = None

    def _step_single_body(self, out):
        g, lib, s = self.geo, self.lib, self.slab
        sc, w, sv = self.sc, self.b, self.Ad          # w = A r lives in b (free once r is formed), s = A d in Ad
        if not (self.fused and self._have_r):
            self._rhs()
            # r = b - A x with gamma = <r, r> (local); then w = A r with delta = <r, w>: ONE all-reduce for the pair
            self._normal(self.x, self.r, self.dots, rhs=self.b, reduce=False)
            sc[0:1].copy_(self.dots[0:1])
        # (one-sweep path: r and gamma were left by the sweep that closed the previous outer iteration)
        if self.cheb:
            if self.fused:      # the sweep that follows reads x and x0 anyway: it returns |x - x0|^2, the solve's last step skips x0
                self._solve_cheb(None)
                self._mark("xsolve")
                self._zu_fused(out[0:1], out[1:2])
            else:
                self._solve_cheb(out[1:2])
                self._zu(out[0:1])
            return
        sc, w, sv = self.sc, self.b, self.Ad          # (b may have been rebound by an earlier Chebyshev solve: take it again)
        self._normal(self.r, w, self.dots, reduce=False)
        sc[1:2].copy_(self.dots[0:1])
        sc[2:4].zero_()                               # alpha_old = 0: first step of this solve
        s.allreduce_sum_(sc[0:2])
        for c in range(self.n_cg):
            last = (c + 1 == self.n_cg)
            _nv.check(lib.tv_cg_update(g.ref, _nv.ptr(self.x), _nv.ptr(self.r), _nv.ptr(self.d), _nv.ptr(sv), _nv.ptr(w),
                                       sc.data_ptr(), _nv.ptr(self.x0) if last else None, out[1:2].data_ptr() if last else None,
                                       _nv.ptr(self.ws), self.stream))
            if last:
                break
            self._normal(self.r, w, self.dots, reduce=False)          # dots = {<r, w>, <r, r>}
            sc[0:1].copy_(self.dots[1:2])
            sc[1:2].copy_(self.dots[0:1])
            s.allreduce_sum_(sc[0:2])
        if self.n_cg == 0:
            out[1:2] = torch.sum((self.x.double() - self.x0.double()) ** 2)     # the slot holds |x - x0|^2: run() halves it
        else:
            out[1:2].mul_(2.0)                        # run() halves: the slot holds |x - x0|^2 like the textbook path
        if self.fused:
            self._zu_fused(out[0:1])
        else:
            self._zu(out[0:1])

    GRAPH_BLOCK = 4             # outer iterations captured per hipGraph
    GRAPH_MAX_VOXELS = 1 << 23  # below this an outer iteration (~10 + 2 n_cg launches) is launch-bound

    def run(self, n_outer, graph=None):
        """n_outer outer iterations; returns the loss history.  graph: None = replay blocks of GRAPH_BLOCK outer iterations
        from a hipGraph when the problem is small enough to be launch-bound and not sharded (every scalar of the CG recurrence
        lives on the device, so an outer iteration has no host round trip to break the capture); True / False force it."""
        hist = torch.zeros((n_outer, 2), dtype=torch.float64, device=self.device)
        use_graph = (self.x0.numel() <= self.GRAPH_MAX_VOXELS) if graph is None else bool(graph)
        start = 0
        if use_graph and not self.slab.sharded and n_outer >= 2 + 2 * self.GRAPH_BLOCK:
            self.step(hist[0])                   # eager: also the warm-up of the capture
            self.step(hist[1])
            start = 2 + self._run_graphed_from(hist, 2, n_outer)
        for k in range(start, n_outer):
            self.step(hist[k])
        self.slab.allreduce_sum_(hist)
        h = hist.cpu().numpy()
        return 0.5 * h[:, 1] + self.reg * h[:, 0]

    def _run_graphed_from(self, hist, first, n_outer):
        """Capture GRAPH_BLOCK outer iterations (not executed during capture) and replay them over hist[first:]."""
        K = self.GRAPH_BLOCK
        nrep = (n_outer - first) // K
        if nrep < 1:
            return 0
        # the Chebyshev x-solve REBINDS x / d / Ad / b at Python level inside every step (the new image is written next to the old
        # one and the roles are swapped): a capture that fails part-way must not leave them pointing at buffers whose kernels
        # never ran (round-3 advice; ChambollePock._run_graphed_from does the same for x / x_alt)
        saved = (self.x, self.d, self.Ad, self.b, self.r, self._have_r)
        saved_u = (self.u, self.u_alt)
        try:
            buf = torch.zeros((K, 2), dtype=torch.float64, device=self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for k in range(K):
                    self.step(buf[k])
        except Exception:
            self.x, self.d, self.Ad, self.b, self.r, self._have_r = saved
            self.u, self.u_alt = saved_u
            return 0                             # nothing ran: undo the bookkeeping, stay eager
        if (self.x is not saved[0]) or (self.d is not saved[1]) or (self.Ad is not saved[2]) or (self.b is not saved[3]) or (self.u is not saved_u[0]):
            # the block does not return the buffers to their roles (odd number of role swaps): replaying it would not be the
            # same iteration twice -- cannot happen with GRAPH_BLOCK even, checked all the same
            self.x, self.d, self.Ad, self.b, self.r, self._have_r = saved
            self.u, self.u_alt = saved_u
            return 0
        done = 0
        for r in range(nrep):
            graph.replay()
            hist[first + done:first + done + K].copy_(buf)
            done += K
        return done

    def result(self):
        return self.x
