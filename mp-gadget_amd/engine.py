"""Host-side mirror of the reference's force-step interface over the C-ABI library (ctypes).

The reference is C; on a machine with the reference toolchain the C-ABI of include/mpgadget_hip.h is bound
by the in-tree shim of INTEGRATION.md.  This module is the same binding for Python callers (tests, bench):
method names, argument meaning and error behaviour follow the reference entry points

    gravpm_init_periodic / gravpm_force          libgadget/gravpm.c:51-119
    force_tree_full / force_tree_rebuild_mask    libgadget/forcetree.c:110-166
    grav_short_tree                              libgadget/gravshort-tree.c:96-154
    set_gravshort_treepar / gravshort_set_softenings / gravshort_fill_ntab / FORCE_SOFTENING

Errors raise EngineError (the reference calls endrun()).  There is no CPU fallback: if the HIP library is
missing or no GPU is present, construction fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmpgadget_hip.so")
TABLE_PATH = os.path.join(_HERE, "data", "shortrange_force_kernels.f64")

# struct particle_data, libgadget/partmanager.h:9-71 (160 bytes)
PARTICLE_DTYPE = np.dtype({
    "names": ["Pos", "TopLeaf", "Mass", "PI", "Flags", "TimeBinHydro", "TimeBinGravity", "Type", "Vel",
              "FullTreeGravAccel", "GravPM", "Ti_drift", "Hsml", "DtHsml", "ID", "GrNr", "Potential"],
    "formats": [("<f8", 3), "<i4", "<f4", "<i4", "u1", "u1", "u1", "u1", ("<f8", 3), ("<f8", 3), ("<f8", 3), "<i8", "<f8",
                "<f8", "<u8", "<i8", "<f8"],
    "offsets": [0, 24, 28, 32, 36, 37, 38, 39, 40, 64, 88, 112, 120, 128, 136, 144, 152],
    "itemsize": 160})

SHORTRANGE_FORCE_WINDOW_TYPE_EXACT = 0   # gravity.h:24-27
SHORTRANGE_FORCE_WINDOW_TYPE_ERFC = 1
ALLMASK, GASMASK, DMMASK, NUMASK, STARMASK, BHMASK = 63, 1, 2, 4, 16, 32   # forcetree.h:22-27


class EngineError(RuntimeError):
    pass


class TreeParams(C.Structure):
    """struct gravshort_tree_params, gravity.h:9-22"""
    _fields_ = [("ErrTolForceAcc", C.c_double), ("BHOpeningAngle", C.c_double), ("MaxBHOpeningAngle", C.c_double),
                ("TreeUseBH", C.c_int), ("Rcut", C.c_double), ("FractionalGravitySoftening", C.c_double)]


class ParticleView(C.Structure):
    _fields_ = [("base", C.c_void_p), ("n", C.c_int64), ("stride", C.c_int64), ("off_pos", C.c_int32),
                ("off_mass", C.c_int32), ("off_flags", C.c_int32), ("off_type", C.c_int32), ("off_accel", C.c_int32),
                ("off_gravpm", C.c_int32), ("off_potential", C.c_int32), ("off_hsml", C.c_int32), ("off_vel", C.c_int32),
                ("off_pi", C.c_int32)]


class TreeStats(C.Structure):
    _fields_ = [("NumParticles", C.c_int64), ("numnodes", C.c_int64), ("numleaves", C.c_int64), ("maxlevel", C.c_int32),
                ("root_mass", C.c_double), ("root_cofm", C.c_double * 3), ("root_hmax", C.c_double)]


class PhaseTimes(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("pm_deposit", "pm_fft", "pm_transfer", "pm_readout", "pm_total", "tree_keys",
                                         "tree_sort", "tree_nodes", "tree_moments", "tree_total", "walk")] + \
               [("walk_launches", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class DensityParams(C.Structure):
    """struct density_params, density.h:10-25"""
    _fields_ = [("DensityResolutionEta", C.c_double), ("MaxNumNgbDeviation", C.c_double), ("BlackHoleNgbFactor", C.c_double),
                ("BlackHoleMaxAccretionRadius", C.c_double), ("DensityKernelType", C.c_int), ("MinGasHsmlFractional", C.c_double)]


class HydroParams(C.Structure):
    """struct hydro_params, hydra.c:26-34"""
    _fields_ = [("DensityIndependentSphOn", C.c_int), ("DensityContrastLimit", C.c_double), ("ArtBulkViscConst", C.c_double)]


class SphTimes(C.Structure):
    """Time-dependent scalars of the SPH loops (kick_factor_data density.h:34-39, drifts hydra.c:178-186, dloga per bin)."""
    _fields_ = [("FgravkickB", C.c_double), ("gravkicks", C.c_double * 47), ("hydrokicks", C.c_double * 47),
                ("drifts", C.c_double * 47), ("dloga_kick", C.c_double * 47), ("dloga_bin", C.c_double * 47),
                ("atime", C.c_double), ("hubble", C.c_double)]


SPH_ARRAY_FIELDS = ("hsml", "dthsml", "vel", "gacc", "gpm", "hydroacc_in", "tb_hydro", "tb_grav", "entropy", "dtentropy_in",
                    "density", "egywtdensity", "dhsmlegyfac", "divvel", "curlvel", "gradrho", "hydroacc_out", "dtentropy_out",
                    "maxsignalvel")


class SphArraysC(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in SPH_ARRAY_FIELDS]


DENSITY_KERNEL_CUBIC_SPLINE, DENSITY_KERNEL_QUINTIC_SPLINE, DENSITY_KERNEL_QUARTIC_SPLINE = 1, 2, 4   # densitykernel.h:17-21

_lib = None


def load_library():
    """dlopen the in-tree HIP library.  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError("HIP engine library %s is missing: run `python __graft_entry__.py` (build()) first; "
                          "there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    # a library that was not built from the sources next to it is refused (build.py: content hash, not file times)
    from . import build as _build
    L.mpg_build_stamp.restype = C.c_char_p
    have = L.mpg_build_stamp().decode()
    if os.path.isdir(_build.CSRC) and not os.environ.get("MPG_ALLOW_STALE_LIBRARY"):
        want = _build.source_stamp()
        if have != want:
            raise EngineError("HIP engine library %s is stale (built from other sources: stamp %s, tree %s): run build() again"
                              % (LIB_PATH, have, want))
    L.mpg_last_error.restype = C.c_char_p
    L.mpg_version.restype = C.c_char_p
    L.mpg_force_softening.restype = C.c_double
    L.mpg_force_softening.argtypes = [C.c_void_p]
    L.mpg_engine_get_stream.restype = C.c_void_p
    L.mpg_engine_get_stream.argtypes = [C.c_void_p]
    L.mpg_engine_destroy.argtypes = [C.c_void_p]
    L.mpg_engine_destroy.restype = None
    L.mpg_dev_tree_order.restype = C.c_void_p
    L.mpg_get_numngb.restype = C.c_double
    L.mpg_get_numngb.argtypes = [C.c_void_p]
    L.mpg_dev_tree_order.argtypes = [C.c_void_p]
    _lib = L
    return L


def _ptr(t):
    """Device pointer of a torch tensor / raw int / None."""
    if t is None:
        return None
    if isinstance(t, int):
        return C.c_void_p(t)
    return C.c_void_p(t.data_ptr())


TIMEBINS = 46


class KickFactors(C.Structure):
    """mpg_kick_factors (include/mpgadget_hip.h)"""
    _fields_ = [("gravkick", C.c_double * (TIMEBINS + 1)), ("hydrokick", C.c_double * (TIMEBINS + 1)), ("dt_entr", C.c_double * (TIMEBINS + 1)),
                ("bin_active", C.c_ubyte * (TIMEBINS + 1)), ("atime", C.c_double), ("MaxGasVel", C.c_double)]


class DriftKickTimes(C.Structure):
    """mpg_drift_kick_times = DriftKickTimes of libgadget/timestep.h:10-27"""
    _fields_ = [("mintimebin", C.c_int), ("maxtimebin", C.c_int), ("mingravtimebin", C.c_int), ("Ti_kick", C.c_int64 * (TIMEBINS + 1)),
                ("Ti_lastactivedrift", C.c_int64 * (TIMEBINS + 1)), ("Ti_Current", C.c_int64), ("PM_length", C.c_int64),
                ("PM_start", C.c_int64), ("PM_kick", C.c_int64)]


class Timeline(C.Structure):
    """mpg_timeline: the sync points of the integer timeline (timebinmgr.c)"""
    _fields_ = [("nsync", C.c_int64), ("loga", C.POINTER(C.c_double))]


class HierGravArrays(C.Structure):
    """mpg_hiergrav_arrays: device arrays of the hierarchical gravity level loop"""
    _fields_ = [("d_vel", C.c_void_p), ("d_gravpm", C.c_void_p), ("d_fulltree_accel", C.c_void_p), ("d_potential", C.c_void_p),
                ("d_tb_grav", C.c_void_p), ("d_flags", C.c_void_p), ("d_stored_accel", C.c_void_p)]


class TimestepParams(C.Structure):
    _fields_ = [("ErrTolIntAccuracy", C.c_double), ("MinSizeTimestep", C.c_double)]


class HydroStepArrays(C.Structure):
    """mpg_hydrostep_arrays: device arrays of find_hydro_timesteps"""
    _fields_ = [(k, C.c_void_p) for k in ("d_type", "d_flags", "d_hsml", "d_dthsml", "d_maxsignalvel", "d_tb_grav", "d_tb_hydro", "d_bh_mintimebin")]


class HydroStepResult(C.Structure):
    """mpg_hydrostep_result"""
    _fields_ = [("mTimeBin", C.c_int), ("ntitype", C.c_int64 * 5), ("badstepsizecount", C.c_int64), ("badtimebins", C.c_int64)]


class TimestepResult(C.Structure):
    """mpg_timestep_result"""
    _fields_ = [("mTimeBin", C.c_int), ("maxTimeBin", C.c_int), ("isPM", C.c_int), ("ntitype", C.c_int64 * 5), ("badstepsizecount", C.c_int64),
                ("badtimebins", C.c_int64)]


class FofParams(C.Structure):
    """mpg_fof_params"""
    _fields_ = [("FOFPrimaryLinkTypes", C.c_int), ("FOFSecondaryLinkTypes", C.c_int), ("FOFHaloComovingLinkingLength", C.c_double),
                ("FOFHaloMinLength", C.c_int)]


class FofGroupsC(C.Structure):
    """mpg_fof_groups"""
    _fields_ = [(k, C.c_void_p) for k in ("MinID", "Length", "GrNr", "LenType", "Mass", "MassType", "CM", "Vel", "Jmom", "Imom", "FirstPos")]


GRAVKICK_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int64, C.c_int64)


class Engine:
    """One engine per GPU (= per MPI rank in the reference's terms).

    Stream semantics of the dev_* (device-resident) calls: they are queued on the ENGINE's stream, which is its own non-blocking
    stream unless use_torch_stream() / set_stream() says otherwise.  Reading their outputs with torch (`.cpu()`, index ops,
    collectives) is ordered after them only if both use one stream - call use_torch_stream() once after construction (bench.py,
    tools/) - or after synchronize().  The host-pointer (AoS) calls synchronise before returning."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.mpg_engine_create(C.byref(h), int(device))
        if rc:
            raise EngineError(self.lib.mpg_last_error().decode())
        self.h = h
        self.device = device
        self._table = None
        self._keep = {}

    # ------------------------------------------------------------------ helpers
    def _ck(self, rc):
        if rc:
            raise EngineError(self.lib.mpg_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.mpg_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def version(self):
        return self.lib.mpg_version().decode()

    def synchronize(self):
        self._ck(self.lib.mpg_engine_synchronize(self.h))

    def set_stream(self, raw_stream):
        self._ck(self.lib.mpg_engine_set_stream(self.h, C.c_void_p(raw_stream)))

    def use_torch_stream(self, device=None):
        """Run the engine and torch (collectives, index ops on the engine's outputs) on ONE stream: a new torch stream is
        made current and handed to the engine.  (torch's default stream is the null stream, whose handle is 0, which
        mpg_engine_set_stream takes as "own non-blocking stream" - work on the two would then be unordered.)"""
        import torch
        torch.cuda.synchronize()   # work already queued on the previous stream (uploads, fills) is done before the switch
        self._torch_stream = torch.cuda.Stream(device=device)
        torch.cuda.set_stream(self._torch_stream)
        self.set_stream(self._torch_stream.cuda_stream)
        return self._torch_stream

    def get_stream(self):
        return self.lib.mpg_engine_get_stream(self.h)

    def set_instrumentation(self, timing=True, counters=False):
        self._ck(self.lib.mpg_set_instrumentation(self.h, int(timing), int(counters)))

    def set_walk_variant(self, variant):
        self._ck(self.lib.mpg_set_walk_variant(self.h, int(variant)))

    def walk_choice(self):
        """(kernel in use, kernel 6 list capacity, targets its last walk handed to the fallback kernel)"""
        v, cap, ovf = C.c_int(0), C.c_int(0), C.c_uint(0)
        self._ck(self.lib.mpg_get_walk_choice(self.h, C.byref(v), C.byref(cap), C.byref(ovf)))
        return v.value, cap.value, ovf.value

    def set_walk_list_capacity(self, cap):
        self._ck(self.lib.mpg_set_walk_list_capacity(self.h, int(cap)))

    def set_walk_split_mode(self, overlap=True, chunks_per_wave=2):
        self._ck(self.lib.mpg_set_walk_split_mode(self.h, int(bool(overlap)), int(chunks_per_wave)))

    def set_walk_offsets64(self, on):
        self._ck(self.lib.mpg_set_walk_offsets64(self.h, int(bool(on))))

    def set_walk_threshold(self, thresh):
        self._ck(self.lib.mpg_set_walk_threshold(self.h, int(thresh)))

    # ------------------------------------------------------------------ module parameters
    def set_gravshort_treepar(self, ErrTolForceAcc=0.002, BHOpeningAngle=0.175, MaxBHOpeningAngle=0.9, TreeUseBH=2,
                              Rcut=6.0, FractionalGravitySoftening=1. / 30.):
        p = TreeParams(ErrTolForceAcc, BHOpeningAngle, MaxBHOpeningAngle, TreeUseBH, Rcut, FractionalGravitySoftening)
        self._ck(self.lib.mpg_set_gravshort_treepar(self.h, C.byref(p)))

    def get_gravshort_treepar(self):
        p = TreeParams()
        self._ck(self.lib.mpg_get_gravshort_treepar(self.h, C.byref(p)))
        return p

    def gravshort_set_softenings(self, MeanSeparation):
        self._ck(self.lib.mpg_gravshort_set_softenings(self.h, C.c_double(MeanSeparation)))

    def FORCE_SOFTENING(self):
        return self.lib.mpg_force_softening(self.h)

    def gravshort_fill_ntab(self, ShortRangeForceWindowType=SHORTRANGE_FORCE_WINDOW_TYPE_EXACT, Asmth=1.5):
        if self._table is None:
            self._table = np.fromfile(TABLE_PATH, dtype="<f8")
            if self._table.size != 512 * 5:
                raise EngineError("corrupt short-range table " + TABLE_PATH)
        self._ck(self.lib.mpg_gravshort_fill_ntab(self.h, int(ShortRangeForceWindowType), C.c_double(Asmth),
                                                  self._table.ctypes.data_as(C.c_void_p), 512))

    def gravpm_init_periodic(self, BoxSize, Asmth, Nmesh, G):
        self._ck(self.lib.mpg_gravpm_init_periodic(self.h, C.c_double(BoxSize), C.c_double(Asmth), int(Nmesh), C.c_double(G)))

    # matter power spectrum of the PM step (gravpm.c:331-382, powerspectrum.c)
    def gravpm_measure_power(self, on=True):
        self._ck(self.lib.mpg_gravpm_measure_power(self.h, int(bool(on))))

    def gravpm_get_powerspectrum(self, nmesh, BoxSize_in_MPC):
        """(kk, Power, Nmodes) of the last PM step in Mpc/h units, empty bins dropped (powerspectrum_sum)."""
        kk, P, N = np.zeros(nmesh), np.zeros(nmesh), np.zeros(nmesh, np.int64)
        nz = C.c_int(0)
        self._ck(self.lib.mpg_gravpm_get_powerspectrum(self.h, C.c_double(BoxSize_in_MPC), kk.ctypes.data_as(C.c_void_p),
                                                       P.ctypes.data_as(C.c_void_p), N.ctypes.data_as(C.c_void_p), C.byref(nz)))
        return kk[:nz.value], P[:nz.value], N[:nz.value]

    def dev_gravpm_powerspectrum_raw(self, acc, modes):
        self._ck(self.lib.mpg_dev_gravpm_powerspectrum_raw(self.h, _ptr(acc), _ptr(modes)))

    def powerspectrum_sum(self, acc, modes, BoxSize_in_MPC):
        """powerspectrum_sum on host arrays acc[2 nbins + 1], modes[nbins] (after the sum over ranks)."""
        acc, modes = np.ascontiguousarray(acc, np.float64), np.ascontiguousarray(modes, np.int64)
        nb = len(modes)
        kk, P, N = np.zeros(nb), np.zeros(nb), np.zeros(nb, np.int64)
        nz = C.c_int(0)
        self._ck(self.lib.mpg_powerspectrum_sum(nb, acc.ctypes.data_as(C.c_void_p), modes.ctypes.data_as(C.c_void_p), C.c_double(BoxSize_in_MPC),
                                                kk.ctypes.data_as(C.c_void_p), P.ctypes.data_as(C.c_void_p), N.ctypes.data_as(C.c_void_p), C.byref(nz)))
        return kk[:nz.value], P[:nz.value], N[:nz.value]

    def powerspectrum_save(self, outdir, filename, Time, D1, kk, P, N):
        kk, P, N = np.ascontiguousarray(kk, np.float64), np.ascontiguousarray(P, np.float64), np.ascontiguousarray(N, np.int64)
        self._ck(self.lib.mpg_powerspectrum_save(outdir.encode(), filename.encode(), C.c_double(Time), C.c_double(D1), len(kk),
                                                 kk.ctypes.data_as(C.c_void_p), P.ctypes.data_as(C.c_void_p), N.ctypes.data_as(C.c_void_p)))

    def set_particle_epoch(self, epoch):
        """Declare the epoch of the host particle table: host calls with the same non-zero epoch, table address, size and box reuse
        the uploaded positions (include/mpgadget_hip.h)."""
        self._ck(self.lib.mpg_set_particle_epoch(self.h, C.c_int64(epoch)))

    def petapm_destroy(self):
        self._ck(self.lib.mpg_petapm_destroy(self.h))

    def init_forcetree_params(self, TreeAllocFactor):
        self._ck(self.lib.mpg_init_forcetree_params(self.h, C.c_double(TreeAllocFactor)))

    # ------------------------------------------------------------------ host (AoS, drop-in) path
    def _view(self, P):
        if P.dtype != PARTICLE_DTYPE or not P.flags["C_CONTIGUOUS"]:
            raise EngineError("P must be a contiguous array of PARTICLE_DTYPE (struct particle_data)")
        v = ParticleView()
        self.lib.mpg_particle_view_reference_layout(C.byref(v), C.c_void_p(P.ctypes.data), C.c_int64(len(P)))
        return v

    # device-resident drop-in mode (include/mpgadget_hip.h): the table P lives in HBM between mpg_resident_begin and _end; the host
    # calls gravpm_force / force_tree_* / grav_short_tree on it move no particle data
    FIELD_POS, FIELD_VEL, FIELD_ACCEL, FIELD_GRAVPM, FIELD_POTENTIAL = 1, 2, 4, 8, 16

    def set_host_overlap(self, on=True):
        """host path: one packing pass per epoch, OldAcc on the device, gravpm_force's results written back while the walk runs
        (mpg_set_host_overlap; results complete when grav_short_tree or host_results_sync returns)"""
        self._ck(self.lib.mpg_set_host_overlap(self.h, int(on)))      # (2 .. 8: that many walk slices whatever the size)

    def host_results_sync(self):
        self._ck(self.lib.mpg_host_results_sync(self.h))

    def resident_begin(self, P, BoxSize):
        v = self._view(P)
        self._ck(self.lib.mpg_resident_begin(self.h, C.byref(v), C.c_double(BoxSize)))

    def resident_fetch(self, P, fields):
        v = self._view(P)
        self._ck(self.lib.mpg_resident_fetch(self.h, C.byref(v), C.c_uint(fields)))

    def resident_push(self, P, fields):
        v = self._view(P)
        self._ck(self.lib.mpg_resident_push(self.h, C.byref(v), C.c_uint(fields)))

    def resident_end(self, P):
        v = self._view(P)
        self._ck(self.lib.mpg_resident_end(self.h, C.byref(v)))

    # a gas run stays resident too: the SPH arrays (host numpy arrays as for density() / hydro_force()) are uploaded once, the two loops and
    # the integrator between them run on the device copies (include/mpgadget_hip.h, mpg_resident_sph_*)
    def resident_sph_begin(self, P, arrays):
        v = self._view(P)
        a = self._sph_host_arrays(arrays)
        self._ck(self.lib.mpg_resident_sph_begin(self.h, C.byref(v), C.byref(a)))

    def resident_sph_end(self, arrays):
        a = self._sph_host_arrays(arrays)
        self._ck(self.lib.mpg_resident_sph_end(self.h, C.byref(a)))

    def resident_drift_all_particles(self, P, ddrift, random_shift=(0.0, 0.0, 0.0)):
        v = self._view(P)
        self._ck(self.lib.mpg_resident_drift_all_particles(self.h, C.byref(v), C.c_double(ddrift), (C.c_double * 3)(*random_shift)))

    def resident_apply_pm_half_kick(self, P, Fgravkick):
        v = self._view(P)
        self._ck(self.lib.mpg_resident_apply_pm_half_kick(self.h, C.byref(v), C.c_double(Fgravkick)))

    def resident_apply_half_kick(self, P, K, ActiveParticle=None):
        v = self._view(P)
        act = None if ActiveParticle is None else np.ascontiguousarray(ActiveParticle, np.int32)
        self._ck(self.lib.mpg_resident_apply_half_kick(self.h, C.byref(v), None if act is None else act.ctypes.data_as(C.c_void_p),
                                                       C.c_int64(0 if act is None else len(act)), C.byref(K)))

    def resident_find_hydro_timesteps(self, P, times, sync_loga, MinSizeTimestep, CourantFac, atime, hubble, ActiveParticle=None, isFirstTimeStep=False):
        v = self._view(P)
        act = None if ActiveParticle is None else np.ascontiguousarray(ActiveParticle, np.int32)
        loga = (C.c_double * len(sync_loga))(*[float(x) for x in sync_loga])
        tl = Timeline(len(sync_loga), C.cast(loga, C.POINTER(C.c_double)))
        par = TimestepParams(0.0, MinSizeTimestep)
        res = HydroStepResult()
        self._ck(self.lib.mpg_resident_find_hydro_timesteps(self.h, C.byref(v), None if act is None else act.ctypes.data_as(C.c_void_p),
                                                            C.c_int64(0 if act is None else len(act)), C.byref(times), C.byref(tl), C.byref(par),
                                                            C.c_double(CourantFac), C.c_double(atime), C.c_double(hubble), int(bool(isFirstTimeStep)),
                                                            C.byref(res)))
        return dict(mTimeBin=res.mTimeBin, ntitype=list(res.ntitype), badstepsizecount=res.badstepsizecount, badtimebins=res.badtimebins)

    def resident_find_timesteps(self, P, times, sync_loga, ErrTolIntAccuracy, MinSizeTimestep, CourantFac, atime, hubble, dti_max_pm=0,
                                ActiveParticle=None):
        """find_timesteps (timestep.c:739-849) on a resident gas run: both time bins of the active particles, times.PM_length on a PM step
        (dti_max_pm = the caller's get_PM_timestep_ti), times.mintimebin / maxtimebin"""
        v = self._view(P)
        act = None if ActiveParticle is None else np.ascontiguousarray(ActiveParticle, np.int32)
        loga = (C.c_double * len(sync_loga))(*[float(x) for x in sync_loga])
        tl = Timeline(len(sync_loga), C.cast(loga, C.POINTER(C.c_double)))
        par = TimestepParams(ErrTolIntAccuracy, MinSizeTimestep)
        res = TimestepResult()
        self._ck(self.lib.mpg_resident_find_timesteps(self.h, C.byref(v), None if act is None else act.ctypes.data_as(C.c_void_p),
                                                      C.c_int64(0 if act is None else len(act)), C.byref(times), C.byref(tl), C.byref(par),
                                                      C.c_double(CourantFac), C.c_double(atime), C.c_double(hubble), C.c_int64(dti_max_pm), C.byref(res)))
        return dict(mTimeBin=res.mTimeBin, maxTimeBin=res.maxTimeBin, isPM=res.isPM, ntitype=list(res.ntitype),
                    badstepsizecount=res.badstepsizecount, badtimebins=res.badtimebins)

    def host_prefetch(self, P, box):
        """mpg_host_prefetch: start this epoch's packing pass + uploads of P[] on a host thread (after set_particle_epoch; needs
        set_host_overlap)"""
        v = self._view(P)
        self._ck(self.lib.mpg_host_prefetch(self.h, C.byref(v), C.c_double(box)))

    def resident_fetch_timebins(self, n):
        """(TimeBinHydro, TimeBinGravity) of a resident gas run as host arrays (what the shim copies into P[] for build_active_particles)"""
        tbh, tbg = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        self._ck(self.lib.mpg_resident_fetch_timebins(self.h, tbh.ctypes.data_as(C.c_void_p), tbg.ctypes.data_as(C.c_void_p)))
        return tbh, tbg

    def resident_arrays(self):
        """device pointers of the resident columns (dict of ints; 0 = absent) and n"""
        class RV(C.Structure):
            _fields_ = [("n", C.c_int64)] + [(k, C.c_void_p) for k in ("d_pos", "d_mass", "d_type", "d_vel", "d_fulltree_accel", "d_gravpm", "d_potential")]
        r = RV()
        self._ck(self.lib.mpg_resident_arrays(self.h, C.byref(r)))
        return {k: (getattr(r, k) or 0) for k, _ in RV._fields_}

    def gravpm_force(self, P):
        v = self._view(P)
        self._ck(self.lib.mpg_gravpm_force(self.h, C.byref(v)))

    def force_tree_full(self, P, BoxSize):
        v = self._view(P)
        self._ck(self.lib.mpg_force_tree_full(self.h, C.byref(v), C.c_double(BoxSize)))

    def force_tree_rebuild_mask(self, P, BoxSize, mask):
        v = self._view(P)
        self._ck(self.lib.mpg_force_tree_rebuild_mask(self.h, C.byref(v), C.c_double(BoxSize), int(mask)))

    def force_tree_free(self):
        self._ck(self.lib.mpg_force_tree_free(self.h))

    def force_tree_active_moments(self, P, BoxSize, ActiveParticle=None, HybridNuTracer=0):
        v = self._view(P)
        act = None if ActiveParticle is None else np.ascontiguousarray(ActiveParticle, np.int32)
        self._ck(self.lib.mpg_force_tree_active_moments(self.h, C.byref(v), C.c_double(BoxSize), None if act is None else act.ctypes.data_as(C.c_void_p),
                                                        C.c_int64(0 if act is None else len(act)), int(HybridNuTracer)))

    def dev_force_tree_active_moments(self, active=None, HybridNuTracer=0):
        self._ck(self.lib.mpg_dev_force_tree_active_moments(self.h, _ptr(active), C.c_int64(0 if active is None else active.shape[0]),
                                                            int(HybridNuTracer)))

    def grav_short_tree(self, P, ActiveParticle=None, AccelStore=None, rho0=0.0):
        v = self._view(P)
        act = None
        nact = 0
        if ActiveParticle is not None:
            act = np.ascontiguousarray(ActiveParticle, np.int32)
            nact = len(act)
        if AccelStore is not None and (AccelStore.dtype != np.float64 or AccelStore.shape != (len(P), 3)
                                       or not AccelStore.flags["C_CONTIGUOUS"]):
            raise EngineError("AccelStore must be a contiguous float64 [NumPart,3] array")
        self._ck(self.lib.mpg_grav_short_tree(self.h, C.byref(v), None if act is None else act.ctypes.data_as(C.c_void_p),
                                              C.c_int64(nact),
                                              None if AccelStore is None else AccelStore.ctypes.data_as(C.c_void_p),
                                              C.c_double(rho0)))

    # ------------------------------------------------------------------ device-resident path (torch tensors on this GPU)
    def dev_bind_particles(self, pos, mass, BoxSize, type=None):
        """pos [n,3] float64, mass [n] float32, type [n] uint8 or None: CUDA(HIP) tensors, contiguous."""
        n = pos.shape[0]
        self._keep["bind"] = (pos, mass, type)
        self._ck(self.lib.mpg_dev_bind_particles(self.h, C.c_int64(n), _ptr(pos), _ptr(mass), _ptr(type), C.c_double(BoxSize)))

    def dev_gravpm_force(self, gravpm, potential=None):
        self._ck(self.lib.mpg_dev_gravpm_force(self.h, _ptr(gravpm), _ptr(potential)))

    def dev_grav_short_pair(self, accel, Rcut, active=None, potential=None, rho0=0.0):
        """grav_short_pair (gravshort-pair.c:21-57) on device-resident arrays"""
        self._ck(self.lib.mpg_dev_grav_short_pair(self.h, _ptr(active), C.c_int64(0 if active is None else active.shape[0]), C.c_double(Rcut),
                                                  _ptr(accel), _ptr(potential), C.c_double(rho0)))

    def grav_short_pair(self, P, Rcut, ActiveParticle=None, rho0=0.0):
        v = self._view(P)
        act = None if ActiveParticle is None else np.ascontiguousarray(ActiveParticle, np.int32)
        self._ck(self.lib.mpg_grav_short_pair(self.h, C.byref(v), None if act is None else act.ctypes.data_as(C.c_void_p),
                                              C.c_int64(0 if act is None else len(act)), C.c_double(Rcut), C.c_double(rho0)))

    # time integration on device-resident arrays (drift.c / timestep.c loops)
    def dev_drift_all_particles(self, pos, vel, ddrift, BoxSize, random_shift=(0.0, 0.0, 0.0), type=None, flags=None, hsml=None, dthsml=None):
        sh = (C.c_double * 3)(*random_shift)
        self._ck(self.lib.mpg_dev_drift_all_particles(self.h, C.c_int64(pos.shape[0]), _ptr(pos), _ptr(vel), _ptr(type), _ptr(flags),
                                                      _ptr(hsml), _ptr(dthsml), C.c_double(ddrift), C.c_double(BoxSize), sh))

    def dev_apply_pm_half_kick(self, vel, gravpm, Fgravkick, flags=None):
        self._ck(self.lib.mpg_dev_apply_pm_half_kick(self.h, C.c_int64(vel.shape[0]), _ptr(vel), _ptr(gravpm), _ptr(flags), C.c_double(Fgravkick)))

    def dev_apply_half_kick(self, vel, gravaccel, K, active=None, type=None, flags=None, tb_grav=None, tb_hydro=None, hydroaccel=None,
                            entropy=None, dtentropy=None):
        self._ck(self.lib.mpg_dev_apply_half_kick(self.h, C.c_int64(vel.shape[0]), _ptr(active), C.c_int64(0 if active is None else active.shape[0]),
                                                  _ptr(vel), _ptr(gravaccel), _ptr(type), _ptr(flags), _ptr(tb_grav), _ptr(tb_hydro),
                                                  _ptr(hydroaccel), _ptr(entropy), _ptr(dtentropy), C.byref(K)))

    def dev_timestep_gravity_dloga(self, gravaccel, gravpm, atime, hubble, ErrTolIntAccuracy, dloga):
        self._ck(self.lib.mpg_dev_timestep_gravity_dloga(self.h, C.c_int64(gravaccel.shape[0]), _ptr(gravaccel), _ptr(gravpm), C.c_double(atime),
                                                         C.c_double(hubble), C.c_double(ErrTolIntAccuracy), _ptr(dloga)))

    def dev_timestep_hydro_dloga(self, type, hsml, dthsml, maxsignalvel, atime, hubble, CourantFac, dloga, titype=None, bh_mintimebin=None,
                                 dloga_for_bin=None):
        """get_timestep_hydro_dloga (timestep.c:1076-1118) for every particle; dloga_for_bin: TIMEBINS + 1 host values or None"""
        tab = None
        if dloga_for_bin is not None:
            tab = (C.c_double * (TIMEBINS + 1))(*[float(x) for x in dloga_for_bin])
        self._ck(self.lib.mpg_dev_timestep_hydro_dloga(self.h, C.c_int64(dloga.shape[0]), _ptr(type), _ptr(hsml), _ptr(dthsml), _ptr(maxsignalvel),
                                                       _ptr(bh_mintimebin), tab, C.c_double(atime), C.c_double(hubble), C.c_double(CourantFac),
                                                       _ptr(dloga), _ptr(titype)))

    def dev_find_hydro_timesteps(self, arrays, active, times, sync_loga, MinSizeTimestep, CourantFac, atime, hubble, isFirstTimeStep=False):
        """find_hydro_timesteps (timestep.c:617-733) on one rank: the particle loop on the device, then the tail that updates
        times.mintimebin (several ranks: all-reduce the result's mTimeBin / counts between the two C-ABI calls).  arrays: dict of device
        tensors type, flags, hsml, dthsml, maxsignalvel, tb_grav, tb_hydro, bh_mintimebin (missing: NULL).  Returns the loop's result."""
        A = HydroStepArrays(*[_ptr(arrays.get(k)) for k in ("type", "flags", "hsml", "dthsml", "maxsignalvel", "tb_grav", "tb_hydro", "bh_mintimebin")])
        loga = (C.c_double * len(sync_loga))(*[float(x) for x in sync_loga])
        tl = Timeline(len(sync_loga), C.cast(loga, C.POINTER(C.c_double)))
        par = TimestepParams(0.0, MinSizeTimestep)
        res = HydroStepResult()
        na = active.shape[0] if active is not None else 0
        self._ck(self.lib.mpg_dev_find_hydro_timesteps(self.h, C.byref(A), _ptr(active), C.c_int64(na), C.byref(times), C.byref(tl), C.byref(par),
                                                       C.c_double(CourantFac), C.c_double(atime), C.c_double(hubble), C.byref(res)))
        n = arrays["tb_hydro"].shape[0]
        self._ck(self.lib.mpg_dev_hydro_timesteps_finish(self.h, int(res.mTimeBin), int(bool(isFirstTimeStep)), C.c_int64(n), _ptr(arrays.get("type")),
                                                         _ptr(arrays.get("tb_hydro")), C.byref(times)))
        return dict(mTimeBin=res.mTimeBin, ntitype=list(res.ntitype), badstepsizecount=res.badstepsizecount, badtimebins=res.badtimebins)

    def dev_find_timesteps(self, arrays, gravaccel, gravpm, active, times, sync_loga, ErrTolIntAccuracy, MinSizeTimestep, CourantFac, atime, hubble,
                           dti_max_pm=0):
        """find_timesteps (timestep.c:739-849) on one rank: the particle loop on the device (both time bins), then the tail (PM step shrink,
        times.mintimebin / maxtimebin).  arrays as dev_find_hydro_timesteps (tb_grav and tb_hydro are both updated)."""
        A = HydroStepArrays(*[_ptr(arrays.get(k)) for k in ("type", "flags", "hsml", "dthsml", "maxsignalvel", "tb_grav", "tb_hydro", "bh_mintimebin")])
        loga = (C.c_double * len(sync_loga))(*[float(x) for x in sync_loga])
        tl = Timeline(len(sync_loga), C.cast(loga, C.POINTER(C.c_double)))
        par = TimestepParams(ErrTolIntAccuracy, MinSizeTimestep)
        res = TimestepResult()
        na = active.shape[0] if active is not None else 0
        self._ck(self.lib.mpg_dev_find_timesteps(self.h, C.byref(A), _ptr(gravaccel), _ptr(gravpm), _ptr(arrays["tb_grav"]), _ptr(active), C.c_int64(na),
                                                 C.byref(times), C.byref(tl), C.byref(par), C.c_double(CourantFac), C.c_double(atime),
                                                 C.c_double(hubble), C.c_int64(dti_max_pm), C.byref(res)))
        self._ck(self.lib.mpg_find_timesteps_finish(int(res.mTimeBin), int(res.maxTimeBin), int(res.isPM), C.byref(times)))
        return dict(mTimeBin=res.mTimeBin, maxTimeBin=res.maxTimeBin, isPM=res.isPM, ntitype=list(res.ntitype),
                    badstepsizecount=res.badstepsizecount, badtimebins=res.badtimebins)

    # particle order: Peano-Hilbert keys and the (type, key) sort (utils/peano.h, slotsmanager.c:404-452)
    def dev_peano_keys(self, pos, box, keys):
        self._ck(self.lib.mpg_dev_peano_keys(self.h, C.c_int64(pos.shape[0]), _ptr(pos), C.c_double(box), _ptr(keys)))

    def dev_order_by_type_and_key(self, keys, perm, type=None, flags=None):
        n = C.c_int64(0)
        self._ck(self.lib.mpg_dev_order_by_type_and_key(self.h, C.c_int64(keys.shape[0]), _ptr(type), _ptr(flags), _ptr(keys), _ptr(perm), C.byref(n)))
        return n.value

    # friends-of-friends groups (fof.c)
    def dev_fof_fof(self, ids, linking_length, min_length=32, vel=None, hsml=None, flags=None, grnr=None, primary=2, secondary=1 + 16 + 32):
        """fof_fof on the bound particles; returns the number of groups.  grnr: int64 [n] device tensor for P[].GrNr (optional)."""
        par = FofParams(int(primary), int(secondary), float(linking_length), int(min_length))
        ng = C.c_int64(0)
        self._ck(self.lib.mpg_dev_fof_fof(self.h, C.byref(par), _ptr(ids), _ptr(vel), _ptr(hsml), _ptr(flags), _ptr(grnr), C.byref(ng)))
        return ng.value

    def dev_fof_groups(self, ngroups, device="cuda"):
        """The group table of the last dev_fof_fof as a dict of device tensors (MinID order)."""
        import torch
        mk = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
        g = dict(MinID=mk(ngroups, torch.int64), Length=mk(ngroups, torch.int32), GrNr=mk(ngroups, torch.int32),
                 LenType=mk((ngroups, 6), torch.int32), Mass=mk(ngroups, torch.float64), MassType=mk((ngroups, 6), torch.float64),
                 CM=mk((ngroups, 3), torch.float64), Vel=mk((ngroups, 3), torch.float64), Jmom=mk((ngroups, 3), torch.float64),
                 Imom=mk((ngroups, 3, 3), torch.float64), FirstPos=mk((ngroups, 3), torch.float32))
        out = FofGroupsC(*[g[k].data_ptr() for k in ("MinID", "Length", "GrNr", "LenType", "Mass", "MassType", "CM", "Vel", "Jmom", "Imom", "FirstPos")])
        self._ck(self.lib.mpg_dev_fof_groups(self.h, C.byref(out)))
        return g

    # hierarchical gravity (timestep.c:239-599)
    @staticmethod
    def _hier_arrays(vel, gravpm, fulltree_accel, tb_grav, potential=None, flags=None, stored_accel=None):
        dp = lambda t: None if t is None else t.data_ptr()
        return HierGravArrays(dp(vel), dp(gravpm), dp(fulltree_accel), dp(potential), dp(tb_grav), dp(flags), dp(stored_accel))

    def dev_build_active_sublist(self, active, tb_grav, flags, maxtimebin, Ti_Current, out):
        n = C.c_int64(0)
        na = active.shape[0] if active is not None else tb_grav.shape[0]
        self._ck(self.lib.mpg_dev_build_active_sublist(self.h, _ptr(active), C.c_int64(na), _ptr(tb_grav), _ptr(flags), int(maxtimebin),
                                                       C.c_int64(Ti_Current), _ptr(out), C.byref(n)))
        return n.value

    def dev_hierarchical_gravity_and_timesteps(self, arrays, active, num_active_gravity, times, sync_loga, ErrTolIntAccuracy, MinSizeTimestep,
                                               atime, hubble, dti_max_pm, rho0, gravkick, HybridNuGrav=0):
        """arrays: HierGravArrays (Engine._hier_arrays); times: DriftKickTimes (updated in place); sync_loga: the sync points' log a;
        gravkick(ti0, ti1) -> get_exact_gravkick_factor.  Returns badstepsizecount."""
        loga = (C.c_double * len(sync_loga))(*[float(x) for x in sync_loga])
        tl = Timeline(len(sync_loga), C.cast(loga, C.POINTER(C.c_double)))
        par = TimestepParams(ErrTolIntAccuracy, MinSizeTimestep)
        fn = GRAVKICK_FN(lambda ctx, a, b: float(gravkick(a, b)))
        bad = C.c_int64(0)
        na = active.shape[0] if active is not None else 0
        nag = num_active_gravity if active is not None else 0
        self._ck(self.lib.mpg_dev_hierarchical_gravity_and_timesteps(self.h, C.byref(arrays), _ptr(active), C.c_int64(na), C.c_int64(nag),
                                                                     C.byref(times), C.byref(tl), C.byref(par), C.c_double(atime),
                                                                     C.c_double(hubble), C.c_int64(dti_max_pm), C.c_double(rho0),
                                                                     int(HybridNuGrav), fn, None, C.byref(bad)))
        return bad.value

    def dev_hierarchical_gravity_accelerations(self, arrays, active, num_active_gravity, times, rho0, gravkick, HybridNuGrav=0):
        fn = GRAVKICK_FN(lambda ctx, a, b: float(gravkick(a, b)))
        na = active.shape[0] if active is not None else 0
        nag = num_active_gravity if active is not None else 0
        self._ck(self.lib.mpg_dev_hierarchical_gravity_accelerations(self.h, C.byref(arrays), _ptr(active), C.c_int64(na), C.c_int64(nag),
                                                                     C.byref(times), C.c_double(rho0), int(HybridNuGrav), fn, None))

    def dev_tree_top_partial(self, La, n_own, out):
        self._ck(self.lib.mpg_dev_tree_top_partial(self.h, int(La), C.c_int64(n_own), _ptr(out)))

    def dev_tree_top_set(self, La, sums):
        self._ck(self.lib.mpg_dev_tree_top_set(self.h, int(La), _ptr(sums)))

    # slab-decomposed PM over several GPUs: local stages (the collectives between them are in pm_slab.py)
    def dev_pm_slab_init(self, rank, world):
        a, b = C.c_int64(0), C.c_int64(0)
        self._ck(self.lib.mpg_dev_pm_slab_init(self.h, int(rank), int(world), C.byref(a), C.byref(b)))
        return a.value, b.value

    def dev_pm_slab_forward_a(self, sendA):
        self._ck(self.lib.mpg_dev_pm_slab_forward_a(self.h, _ptr(sendA)))

    def dev_pm_slab_forward_b(self, recvA, sendB):
        self._ck(self.lib.mpg_dev_pm_slab_forward_b(self.h, _ptr(recvA), _ptr(sendB)))

    def dev_pm_slab_inverse_c(self, recvB, ghost_send):
        self._ck(self.lib.mpg_dev_pm_slab_inverse_c(self.h, _ptr(recvB), _ptr(ghost_send)))

    def dev_pm_slab_readout(self, ghost_recv, targets, gravpm, potential=None):
        self._ck(self.lib.mpg_dev_pm_slab_readout(self.h, _ptr(ghost_recv), _ptr(targets), C.c_int64(targets.shape[0]), _ptr(gravpm),
                                                  _ptr(potential)))

    def dev_force_tree_build(self, mask=ALLMASK):
        self._ck(self.lib.mpg_dev_force_tree_build(self.h, int(mask)))

    def dev_grav_short_tree(self, accel, oldacc=None, prev_accel=None, gravpm=None, active=None, potential=None, rho0=0.0,
                            nactive=None):
        """active: int32 device tensor of caller indices, or a raw device pointer (int) with `nactive` entries."""
        if active is None:
            nact = 0
        elif nactive is not None:
            nact = int(nactive)
        else:
            nact = active.shape[0]
        self._ck(self.lib.mpg_dev_grav_short_tree(self.h, _ptr(oldacc), _ptr(prev_accel), _ptr(gravpm), _ptr(active),
                                                  C.c_int64(nact), _ptr(accel), _ptr(potential), C.c_double(rho0)))

    # ------------------------------------------------------------------ SPH (device-resident)
    def set_densitypar(self, DensityResolutionEta=1.0, MaxNumNgbDeviation=2.0, BlackHoleNgbFactor=2.0,
                       BlackHoleMaxAccretionRadius=99999., DensityKernelType=DENSITY_KERNEL_QUINTIC_SPLINE,
                       MinGasHsmlFractional=0.006):
        p = DensityParams(DensityResolutionEta, MaxNumNgbDeviation, BlackHoleNgbFactor, BlackHoleMaxAccretionRadius,
                          DensityKernelType, MinGasHsmlFractional)
        self._ck(self.lib.mpg_set_densitypar(self.h, C.byref(p)))

    def set_hydropar(self, DensityIndependentSphOn=1, DensityContrastLimit=100.0, ArtBulkViscConst=0.75):
        p = HydroParams(DensityIndependentSphOn, DensityContrastLimit, ArtBulkViscConst)
        self._ck(self.lib.mpg_set_hydropar(self.h, C.byref(p)))

    def GetNumNgb(self):
        return self.lib.mpg_get_numngb(self.h)

    @staticmethod
    def _sph_arrays(arrays):
        """arrays: dict name -> device tensor (torch) / raw pointer / None for the fields of mpg_sph_arrays."""
        a = SphArraysC()
        for k in SPH_ARRAY_FIELDS:
            t = arrays.get(k)
            setattr(a, k, None if t is None else (t if isinstance(t, int) else t.data_ptr()))
        return a

    def dev_force_tree_rebuild_mask(self, mask, with_moments=False):
        self._ck(self.lib.mpg_dev_force_tree_rebuild_mask(self.h, int(mask), int(bool(with_moments))))

    def dev_set_init_hsml(self, arrays, MeanGasSeparation):
        a = self._sph_arrays(arrays)
        self._ck(self.lib.mpg_dev_set_init_hsml(self.h, C.byref(a), C.c_double(MeanGasSeparation)))

    def dev_density(self, arrays, times, active=None, update_hsml=1, DoEgyDensity=0, BlackHoleOn=0):
        a = self._sph_arrays(arrays)
        nact = 0 if active is None else active.shape[0]
        self._ck(self.lib.mpg_dev_density(self.h, C.byref(a), C.byref(times), _ptr(active), C.c_int64(nact), int(update_hsml),
                                          int(DoEgyDensity), int(BlackHoleOn)))

    def dev_force_tree_calc_hmax(self, hsml=None):
        """hmax moments of the gas tree: from the arrays of the last density() (run.c:477), or - force_update_hmax - from `hsml`"""
        if hsml is None:
            self._ck(self.lib.mpg_dev_force_tree_calc_hmax(self.h))
        else:
            self._keep["hmax_hsml"] = hsml
            self.lib.mpg_dev_force_update_hmax.argtypes = [C.c_void_p, C.c_void_p]
            self._ck(self.lib.mpg_dev_force_update_hmax(self.h, C.c_void_p(hsml.data_ptr())))

    def dev_hydro_force(self, arrays, times, active=None):
        a = self._sph_arrays(arrays)
        nact = 0 if active is None else active.shape[0]
        self._ck(self.lib.mpg_dev_hydro_force(self.h, C.byref(a), C.byref(times), _ptr(active), C.c_int64(nact)))

    # host-pointer SPH path: `arrays` maps field names of mpg_sph_arrays to contiguous numpy arrays in particle order
    @staticmethod
    def _sph_host_arrays(arrays):
        a = SphArraysC()
        for k in SPH_ARRAY_FIELDS:
            t = arrays.get(k)
            if t is not None:
                want = np.uint8 if k.startswith("tb_") else np.float64
                if t.dtype != want or not t.flags["C_CONTIGUOUS"]:
                    raise EngineError("SPH host array %s must be contiguous %s" % (k, want.__name__))
            setattr(a, k, None if t is None else t.ctypes.data)
        return a

    def set_init_hsml(self, P, BoxSize, arrays, MeanGasSeparation):
        v = self._view(P)
        a = self._sph_host_arrays(arrays)
        self._ck(self.lib.mpg_set_init_hsml(self.h, C.byref(v), C.c_double(BoxSize), C.byref(a), C.c_double(MeanGasSeparation)))

    def density(self, P, BoxSize, arrays, times, ActiveParticle=None, update_hsml=1, DoEgyDensity=0, BlackHoleOn=0):
        v = self._view(P)
        a = self._sph_host_arrays(arrays)
        act = None if ActiveParticle is None else np.ascontiguousarray(ActiveParticle, np.int32)
        self._ck(self.lib.mpg_density(self.h, C.byref(v), C.c_double(BoxSize), C.byref(a), C.byref(times),
                                      None if act is None else act.ctypes.data_as(C.c_void_p), C.c_int64(0 if act is None else len(act)),
                                      int(update_hsml), int(DoEgyDensity), int(BlackHoleOn)))

    def hydro_force(self, P, arrays, times, ActiveParticle=None):
        v = self._view(P)
        a = self._sph_host_arrays(arrays)
        act = None if ActiveParticle is None else np.ascontiguousarray(ActiveParticle, np.int32)
        self._ck(self.lib.mpg_hydro_force(self.h, C.byref(v), C.byref(a), C.byref(times),
                                          None if act is None else act.ctypes.data_as(C.c_void_p), C.c_int64(0 if act is None else len(act))))

    def sph_stats(self):
        c = (C.c_int64 * 4)()
        self._ck(self.lib.mpg_sph_get_stats(self.h, c))
        return dict(iterations=c[0], targets=c[1], interactions=c[2], candidates=c[3])

    # ------------------------------------------------------------------ introspection
    def tree_stats(self):
        st = TreeStats()
        self._ck(self.lib.mpg_tree_get_stats(self.h, C.byref(st)))
        return st

    def tree_export(self):
        st = self.tree_stats()
        n = st.numnodes
        d = dict(level=np.zeros(n, np.int32), center=np.zeros((n, 3)), len=np.zeros(n), cofm=np.zeros((n, 3)),
                 mass=np.zeros(n), hmax=np.zeros(n), sibling=np.zeros(n, np.int32), pstart=np.zeros(n, np.int32),
                 pcount=np.zeros(n, np.int32))
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._ck(self.lib.mpg_tree_export(self.h, p(d["level"]), p(d["center"]), p(d["len"]), p(d["cofm"]), p(d["mass"]),
                                          p(d["hmax"]), p(d["sibling"]), p(d["pstart"]), p(d["pcount"])))
        order = np.zeros(st.NumParticles, np.int32)
        self._ck(self.lib.mpg_tree_export_order(self.h, order.ctypes.data_as(C.c_void_p)))
        d["order"] = order
        return d

    def walk_counters(self):
        c = (C.c_int64 * 10)()
        self._ck(self.lib.mpg_walk_get_counters(self.h, c))
        return dict(pp=c[0], nodes_visited=c[1], nodes_used=c[2], targets=c[3], node_steps=c[4], node_lanes=c[5],
                    int_steps=c[6], int_lanes=c[7], cycles_a=c[8], cycles_b=c[9])

    def walk_f32_stats(self):
        """(fp64 fall-back passes, waves on the fp32 node tests) of the last counting walk with MPG_LISTS_F32=1"""
        c = (C.c_int64 * 2)()
        self._ck(self.lib.mpg_walk_get_f32_stats(self.h, c))
        return int(c[0]), int(c[1])

    def walk_events_collect(self):
        """(total_ms, launches) of the walk kernel since the last collect, from HIP events on the engine stream."""
        tot = C.c_double()
        cnt = C.c_int()
        self._ck(self.lib.mpg_walk_events_collect(self.h, C.byref(tot), C.byref(cnt)))
        return tot.value, cnt.value

    def walk_events_collect_split(self):
        """(total_ms, launches, lists_ms, eval_ms, split_launches): as walk_events_collect, with the two kernels' times of the walks that
        ran as one list kernel + one evaluation kernel."""
        tot, tl, te = C.c_double(), C.c_double(), C.c_double()
        cnt, cs = C.c_int(), C.c_int()
        self._ck(self.lib.mpg_walk_events_collect2(self.h, C.byref(tot), C.byref(cnt), C.byref(tl), C.byref(te), C.byref(cs)))
        return tot.value, cnt.value, tl.value, te.value, cs.value

    def dev_tree_order(self, n, device):
        """Zero-copy int32 torch view of the engine-owned tree-order permutation (tree slot -> caller index)."""
        import torch

        class _Holder:
            pass
        h = _Holder()
        h.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i4", "data": (int(self.dev_tree_order_ptr()), False), "version": 2}
        return torch.as_tensor(h, device=device)

    def dev_tree_order_ptr(self):
        """Raw device pointer (int) of the tree-order permutation, int32 [NumParticles]."""
        return self.lib.mpg_dev_tree_order(self.h)

    def phase_times(self):
        t = PhaseTimes()
        self._ck(self.lib.mpg_get_phase_times(self.h, C.byref(t)))
        return t.as_dict()


def make_particles(pos, mass, type=1):
    """Build a struct particle_data table (PARTICLE_DTYPE) from positions and masses (test/bench helper)."""
    P = np.zeros(len(pos), dtype=PARTICLE_DTYPE)
    P["Pos"] = pos
    P["Mass"] = mass
    P["Type"] = type
    P["ID"] = np.arange(len(pos), dtype=np.uint64)
    return P
