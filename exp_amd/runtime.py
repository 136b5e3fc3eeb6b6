"""Host-side mirror of EXP's force-method interface over the C ABI.

Class and method names follow the reference so that callers read the same:

* ``Component``       <- ``Component`` particle accessors / CUDA mirror
                         (``src/Component.H:738-760``, ``src/cudaComponent.cu:621-727``)
* ``SphereSL``        <- ``Sphere : SphericalBasis : PotAccel``
                         (``src/Sphere.cc:28-96``, ``src/PotAccel.H:173-288``)
* ``incr_position`` / ``incr_velocity`` <- ``src/incpos.cc:72``, ``src/incvel.cc:90``

Everything numeric happens in libexp_amd.so on the GPU; this file only marshals
arguments (numpy host arrays or torch device tensors) and keeps object lifetimes.
"""
from __future__ import annotations

import ctypes
from ctypes import byref, c_char_p, c_double, c_int, c_longlong, c_void_p
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import CylConfig, SphConfig, as_f64, check
from .empcyl import EmpCylGrid
from .slgrid import SLGridSph


class Context:
    """One GPU + one HIP stream (``exp_amd_ctx``)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = _lib.load()
        h = c_void_p()
        check(self.lib.exp_amd_ctx_create(int(device), c_void_p(stream) if stream else None,
                                          byref(h)))
        self.h = h
        self.device = device
        self._cb = None
        self._children = []

    # -- collectives ---------------------------------------------------------------------
    def set_allreduce(self, fn, nranks: Optional[int] = None, rank: int = 0) -> None:
        """fn(ptr:int, count:int, stream:int) reduces `count` doubles in place (SUM) over `nranks` ranks (the size
        and rank of the host's communicator, ``exp_amd_comm_set_world``: what ``comm_info`` reports; optional)."""
        if nranks is not None:
            check(self.lib.exp_amd_comm_set_world(self.h, int(nranks), int(rank)), self.h)
        if fn is None:
            self._cb = None
            check(self.lib.exp_amd_comm_set_callback(self.h, _lib.ALLREDUCE_FN(), None), self.h)
            return

        def tramp(buf, count, stream, user):
            try:
                fn(int(buf), int(count), int(stream or 0))
                return 0
            except Exception as e:  # pragma: no cover - surfaced through the C error path
                import traceback
                traceback.print_exc()
                return 1

        self._cb = _lib.ALLREDUCE_FN(tramp)
        check(self.lib.exp_amd_comm_set_callback(self.h, self._cb, None), self.h)

    def init_rccl(self, unique_id: bytes, nranks: int, rank: int) -> None:
        buf = ctypes.create_string_buffer(unique_id, 128)
        check(self.lib.exp_amd_comm_init_rank(self.h, buf, nranks, rank), self.h)

    @staticmethod
    def rccl_unique_id() -> bytes:
        lib = _lib.load()
        buf = ctypes.create_string_buffer(128)
        check(lib.exp_amd_comm_get_unique_id(buf))
        return buf.raw

    def comm_info(self) -> dict:
        """Which coefficient all-reduce this context uses and how often it ran."""
        from ctypes import c_longlong
        kind, nr, rk, calls = c_int(), c_int(), c_int(), c_longlong()
        check(self.lib.exp_amd_comm_info(self.h, byref(kind), byref(nr), byref(rk), byref(calls)), self.h)
        return {"kind": ("none", "rccl", "callback")[kind.value], "nranks": nr.value, "rank": rk.value,
                "allreduce_calls": int(calls.value), "streams": int(self.lib.exp_amd_comm_streams(self.h))}

    def allreduce(self, device_ptr: int, count: int) -> None:
        """In-place sum over ranks of ``count`` doubles at a device pointer, on the context's stream."""
        check(self.lib.exp_amd_comm_allreduce(self.h, c_void_p(int(device_ptr)), int(count)), self.h)

    def allreduce_max(self, value: float) -> float:
        """MAX over the ranks of one host number, on the context's own transport (``exp_amd_comm_allreduce_max``)."""
        v = c_double(float(value))
        check(self.lib.exp_amd_comm_allreduce_max(self.h, byref(v)), self.h)
        return float(v.value)

    def set_deterministic(self, on: bool = True) -> None:
        """Order-independent coefficient sums: runs become bit-reproducible (include/exp_amd.h)."""
        check(self.lib.exp_amd_ctx_set_deterministic(self.h, int(bool(on))), self.h)

    def set_prekick(self, on: bool = True) -> None:
        """Fused steps store velocities with the next opening half-kick applied (include/exp_amd.h)."""
        check(self.lib.exp_amd_ctx_set_prekick(self.h, int(bool(on))), self.h)

    def set_dense_min(self, nmin: int) -> None:
        """Block multistep: levels with fewer particles than this are not cell-sorted (0: all are)."""
        check(self.lib.exp_amd_ctx_set_dense_min(self.h, int(nmin)), self.h)

    def set_thin_max(self, nmax: int) -> None:
        """Block multistep: active slot ranges of at most this many particles (all in sparse levels) are accumulated and
        evaluated straight from the basis tables, without moments or a projected table (0: never)."""
        check(self.lib.exp_amd_ctx_set_thin_max(self.h, int(nmax)), self.h)

    def set_mover_list_min(self, nmin: int) -> None:
        """Block multistep: from this many level changes in a sweep on, the coefficient differencing runs the list of
        movers through the accumulation kernels instead of per-particle atomics (0: always, < 0: never)."""
        check(self.lib.exp_amd_ctx_set_mover_list_min(self.h, int(nmin)), self.h)

    def set_append_min(self, nmin: int) -> None:
        """Single-level components of at least ``nmin`` particles take the APPEND fused step (no sort passes: the force pass
        places every particle in the next step's cell order; include/exp_amd.h).  ``nmin <= 0``: never."""
        check(self.lib.exp_amd_ctx_set_append_min(self.h, int(nmin)), self.h)

    def set_append_lean(self, on: bool) -> None:
        """The append step with the LEAN payload: the placing pass stores neither acceleration nor potential; the first call
        that looks at the component has them re-evaluated from the coefficient set kept at the completed step
        (include/exp_amd.h: exp_amd_ctx_set_append_lean).  Off by default."""
        check(self.lib.exp_amd_ctx_set_append_lean(self.h, int(bool(on))), self.h)

    def set_split_min(self, nmin: int) -> None:
        """Smallest component the fused step handles as two overlapped halves (<= 0: never)."""
        check(self.lib.exp_amd_ctx_set_split_min(self.h, int(nmin)), self.h)

    def synchronize(self) -> None:
        check(self.lib.exp_amd_ctx_synchronize(self.h), self.h)

    @property
    def stream(self) -> int:
        return int(self.lib.exp_amd_ctx_stream(self.h) or 0)

    # -- profiling -----------------------------------------------------------------------
    def profile(self, on: bool = True) -> None:
        check(self.lib.exp_amd_profile_enable(self.h, int(on)), self.h)

    def profile_reset(self) -> None:
        check(self.lib.exp_amd_profile_reset(self.h), self.h)

    def profile_report(self) -> dict:
        out = {}
        i = 0
        while True:
            name = c_char_p()
            ms = c_double()
            cnt = c_longlong()
            if self.lib.exp_amd_profile_get(self.h, i, byref(name), byref(ms), byref(cnt)) != 0:
                break
            out[name.value.decode()] = {"ms_total": ms.value, "launches": cnt.value}
            i += 1
        return out

    def close(self) -> None:
        if self.h:
            for ch in list(self._children):
                ch.close()
            self.lib.exp_amd_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Component:
    """Particle store of one component (fp64 SoA in HBM)."""

    def __init__(self, ctx: Context, n: int):
        self.ctx, self.lib, self.n = ctx, ctx.lib, int(n)
        h = c_void_p()
        check(self.lib.exp_amd_comp_create(ctx.h, self.n, byref(h)), ctx.h)
        self.h = h
        ctx._children.append(self)

    @classmethod
    def from_arrays(cls, ctx: Context, mass, pos, vel=None) -> "Component":
        pos = np.asarray(pos, dtype=np.float64)
        c = cls(ctx, pos.shape[0])
        c.upload(mass, pos, vel)
        return c

    @classmethod
    def from_frame(cls, ctx: Context, mass, pos, vel=None, center=None, rot=None) -> "Component":
        """Particles as the caller holds them ([n, 3], C- or Fortran-ordered), taken into the expansion frame
        ``rot (x - center)`` on the device (``exp_amd_comp_upload_frame``): no host-side column copies, no host product."""
        pos = np.asarray(pos, dtype=np.float64)
        c = cls(ctx, pos.shape[0])
        c.upload_frame(mass, pos, vel, center, rot)
        return c

    def upload_frame(self, mass, pos, vel=None, center=None, rot=None) -> None:
        def parts(a):
            """-> (x, y, z pointers, stride, keep-alive)"""
            if a is None:
                return (None, None, None), 1, None
            a = np.asarray(a, dtype=np.float64)
            assert a.shape == (self.n, 3)
            if a.T.flags.c_contiguous and self.n > 1:           # three contiguous columns (a [3, n] array seen as [n, 3])
                return tuple(a[:, k].ctypes.data_as(c_void_p) for k in range(3)), 1, a
            a = np.ascontiguousarray(a)
            return (a.ctypes.data_as(c_void_p), None, None), 3, a
        (x, y, z), stride, keep_p = parts(pos)
        (vx, vy, vz), vstride, keep_v = parts(vel)
        if vel is not None and vstride != stride:               # one stride per call: make the velocities match
            v = np.asarray(vel, dtype=np.float64)
            keep_v = np.ascontiguousarray(v) if stride == 3 else np.asfortranarray(v)
            (vx, vy, vz), vstride, keep_v = parts(keep_v)
        m = as_f64(mass, self.n)
        ctr = None if center is None else np.ascontiguousarray(center, dtype=np.float64).reshape(3)
        R = None if rot is None else np.ascontiguousarray(rot, dtype=np.float64).reshape(9)
        check(self.lib.exp_amd_comp_upload_frame(self.h, m[1], x, y, z, vx, vy, vz, int(stride),
                                                 None if ctr is None else ctr.ctypes.data_as(c_void_p),
                                                 None if R is None else R.ctypes.data_as(c_void_p)), self.ctx.h)
        del keep_p, keep_v

    def upload(self, mass, pos, vel=None) -> None:
        pos = np.asarray(pos, dtype=np.float64)
        keep = [as_f64(mass, self.n)] + [as_f64(pos[:, k], self.n) for k in range(3)]
        if vel is not None:
            vel = np.asarray(vel, dtype=np.float64)
            keep += [as_f64(vel[:, k], self.n) for k in range(3)]
        else:
            keep += [(None, None)] * 3
        p = [k[1] for k in keep]
        check(self.lib.exp_amd_comp_upload(self.h, *p), self.ctx.h)

    def upload_device(self, mass, x, y, z, vx=None, vy=None, vz=None) -> None:
        """Adopt torch CUDA tensors (float64, contiguous, length n) already in HBM."""
        ptrs = []
        for t in (mass, x, y, z, vx, vy, vz):
            if t is None:
                ptrs.append(None)
                continue
            assert t.is_cuda and t.is_contiguous() and t.numel() == self.n and t.element_size() == 8
            ptrs.append(c_void_p(t.data_ptr()))
        import torch
        torch.cuda.current_stream().synchronize()
        check(self.lib.exp_amd_comp_upload_device(self.h, *ptrs), self.ctx.h)

    def upload_acc(self, acc, pot=None) -> None:
        acc = np.asarray(acc, dtype=np.float64)
        keep = [as_f64(acc[:, k], self.n) for k in range(3)] + [as_f64(pot, self.n)]
        check(self.lib.exp_amd_comp_upload_acc(self.h, *[k[1] for k in keep]), self.ctx.h)

    def upload_levels(self, level) -> None:
        lv = np.ascontiguousarray(level, dtype=np.int32)
        assert lv.size == self.n
        check(self.lib.exp_amd_comp_upload_levels(self.h, lv.ctypes.data_as(c_void_p)), self.ctx.h)

    def download(self, fields: Sequence[str] = ("mass", "pos", "vel", "acc", "pot")) -> dict:
        n = self.n
        bufs = {k: np.empty(n) for k in ("mass", "x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az",
                                         "pot")}
        want = set()
        if "mass" in fields: want |= {"mass"}
        if "pos" in fields: want |= {"x", "y", "z"}
        if "vel" in fields: want |= {"vx", "vy", "vz"}
        if "acc" in fields: want |= {"ax", "ay", "az"}
        if "pot" in fields: want |= {"pot"}
        order = ("mass", "x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az", "pot")
        ptrs = [bufs[k].ctypes.data_as(c_void_p) if k in want else None for k in order]
        check(self.lib.exp_amd_comp_download(self.h, *ptrs), self.ctx.h)
        out = {}
        if "mass" in fields: out["mass"] = bufs["mass"]
        if "pos" in fields: out["pos"] = np.stack([bufs["x"], bufs["y"], bufs["z"]], 1)
        if "vel" in fields: out["vel"] = np.stack([bufs["vx"], bufs["vy"], bufs["vz"]], 1)
        if "acc" in fields: out["acc"] = np.stack([bufs["ax"], bufs["ay"], bufs["az"]], 1)
        if "pot" in fields: out["pot"] = bufs["pot"]
        return out

    def download_levels(self) -> np.ndarray:
        lv = np.empty(self.n, dtype=np.int32)
        check(self.lib.exp_amd_comp_download_levels(self.h, lv.ctypes.data_as(c_void_p)), self.ctx.h)
        return lv

    def set_rtrunc(self, rtrunc: float, com0=None) -> None:
        """The component's ``rtrunc`` key with the ``com0`` it is measured from (src/Component.cc:69, :213, :1023):
        ``Component::freeze`` (:4194-4202) -- a particle with |pos - com0 - center| > rtrunc takes no part in any force
        method's accumulation, level-change differencing or force pass."""
        c0 = None if com0 is None else as_f64(np.asarray(com0, dtype=np.float64).reshape(3))
        check(self.lib.exp_amd_comp_set_rtrunc(self.h, float(rtrunc), c0[1] if c0 else None), self.ctx.h)
        self.rtrunc = float(rtrunc)
        self.com0 = None if com0 is None else np.asarray(com0, dtype=np.float64).reshape(3).copy()

    def set_level_policy(self, noswitch: bool = False, freeze_levels: bool = False, dtreset: bool = True) -> None:
        """The component keys ``noswitch``, ``freezeL``, ``dtreset`` (src/Component.cc:253-255, :1036-1038) that
        ``adjust_multistep_level`` reads (src/multistep.cc:136-158, :528-534).  ``freezeL``: levels are assigned on the first
        call only.  ``noswitch``: ``Particle::dtreq`` keeps the smallest time step asked for since its last reset (at mstep ==
        0 with ``dtreset``, and on the first call) and levels are assigned at the end of a master step only."""
        check(self.lib.exp_amd_comp_set_level_policy(self.h, int(bool(noswitch)), int(bool(freeze_levels)), int(bool(dtreset))),
              self.ctx.h)

    def set_consp(self, rcom: float, on: bool = True) -> None:
        """The component keys ``tidal`` (switches ``consp`` on, src/Component.cc:998-1000) and ``rcom`` (:1024): from now on
        ``fix_positions`` flags a particle beyond ``rcom`` of com0 + center (``escape_com``, :4204-4212) and leaves it out of
        the centre-of-mass sums for good (:3317-3334).  ``escaped()`` is the attribute column ``iattrib[tidal]``."""
        check(self.lib.exp_amd_comp_set_consp(self.h, int(bool(on)), float(rcom)), self.ctx.h)
        self.rcom = float(rcom)

    def escaped(self) -> np.ndarray:
        fl = np.empty(self.n, dtype=np.uint8)
        check(self.lib.exp_amd_comp_get_escaped(self.h, fl.ctypes.data_as(c_void_p)), self.ctx.h)
        return fl

    def set_escaped(self, flags) -> None:
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        if fl.shape != (self.n,):
            raise ValueError("set_escaped: one flag per particle")
        check(self.lib.exp_amd_comp_set_escaped(self.h, fl.ctypes.data_as(c_void_p)), self.ctx.h)

    # Component::Adiabatic() (src/Component.cc:4214-4220; keys ton, toff, twid :1040-1055)
    adiabatic = None

    def set_adiabatic(self, ton: float = -1.0e20, toff: float = 1.0e20, twid: float = 0.1) -> None:
        self.adiabatic = (float(ton), float(toff), float(twid))

    def Adiabatic(self, tnow: float) -> float:
        if self.adiabatic is None:
            return 1.0
        import math
        ton, toff, twid = self.adiabatic
        return 0.25 * (1.0 + math.erf((tnow - ton) / twid)) * (1.0 + math.erf((toff - tnow) / twid))

    def set_center(self, center) -> None:
        c = (c_double * 3)(*[float(v) for v in center])
        check(self.lib.exp_amd_comp_set_center(self.h, c), self.ctx.h)

    @property
    def center(self) -> np.ndarray:
        """``Component::center`` as it stands (the caller's, or the orientation estimator's inside a sim)."""
        c = (c_double * 3)()
        check(self.lib.exp_amd_comp_get_center(self.h, c), self.ctx.h)
        return np.array(c[:])

    # src/incpos.cc:72, src/incvel.cc:90
    def set_orientation(self, body=None) -> None:
        """Body-frame rotation of the component (``Orient::transformBody``), used by the cylindrical
        force method after centring (src/Cylinder.cc:799, :1352, :1418); ``None`` removes it."""
        if body is None:
            check(self.lib.exp_amd_comp_set_orientation(self.h, None), self.ctx.h)
        else:
            b = np.ascontiguousarray(body, dtype=np.float64).reshape(9)
            check(self.lib.exp_amd_comp_set_orientation(self.h, b.ctypes.data_as(c_void_p)), self.ctx.h)

    def set_pseudo_accel(self, accel=None, omega=None, domdt=None) -> None:
        """Frame acceleration subtracted by every force applied to this component
        (``Component::AddAcc`` / ``getPseudoAccel``, src/Component.H:914-921, src/Component.cc:4407-4427)."""
        arrs = [None if v is None else np.ascontiguousarray(v, dtype=np.float64).reshape(3)
                for v in (accel, omega, domdt)]
        ptr = [None if a is None else a.ctypes.data_as(c_void_p) for a in arrs]
        check(self.lib.exp_amd_comp_set_pseudo_accel(self.h, *ptr), self.ctx.h)

    def incr_position(self, dt: float, mlevel: int = -1) -> None:
        check(self.lib.exp_amd_comp_drift(self.h, float(dt), int(mlevel)), self.ctx.h)

    def incr_velocity(self, dt: float, mlevel: int = -1) -> None:
        check(self.lib.exp_amd_comp_kick(self.h, float(dt), int(mlevel)), self.ctx.h)

    def zero_acceleration(self, mlevel: int = 0) -> None:
        check(self.lib.exp_amd_comp_zero_acc(self.h, int(mlevel)), self.ctx.h)

    def fix_positions(self, mlevel: int = 0) -> dict:
        """``Component::fix_positions`` (src/Component.cc:3280-3554): total mass and the centres
        of mass, velocity and acceleration; levels below ``mlevel`` keep their last sums."""
        out = (c_double * 10)()
        check(self.lib.exp_amd_comp_fix_positions(self.h, int(mlevel), out), self.ctx.h)
        v = np.array(out[:])
        return {"mtot": float(v[0]), "com": v[1:4].copy(), "cov": v[4:7].copy(), "coa": v[7:10].copy()}

    def log_sums(self) -> dict:
        """The sums ``OutLog::Run`` forms over a component (src/OutLog.cc:392-478), on the device."""
        out = (c_double * 14)()
        check(self.lib.exp_amd_comp_log_sums(self.h, out), self.ctx.h)
        v = np.array(out[:])
        return {"mtot": float(v[0]), "com": v[1:4].copy(), "cov": v[4:7].copy(), "angm": v[7:10].copy(),
                "ektot": float(v[10]), "eptot": float(v[11]), "clausius": float(v[12]), "nbodies": int(round(v[13]))}

    def close(self) -> None:
        if self.h:
            self.lib.exp_amd_comp_destroy(self.h)
            self.h = None
            if self in self.ctx._children:
                self.ctx._children.remove(self)


class Orient:
    """``Orient`` (src/Orient.H:31-204): the most-bound-particle estimator of a component's
    expansion centre and symmetry axis.  Constructor arguments are the reference's
    (number_to_keep, target, orient flags, control flags, dT, damping); ``Naccel`` is ``set_naccel``
    and ``Logfile`` with its restart is ``openLog``."""

    AXIS, CENTER = 1, 2                     # Orient::OrientFlags
    DIAG, KE, EXTERNAL = 1, 2, 4            # Orient::ControlFlags

    def __init__(self, ctx: Context, number_to_keep: int, target: int, orient_flags: int,
                 control_flags: int = 0, dT: float = 0.0, damping: float = 1.0):
        self.ctx, self.lib = ctx, ctx.lib
        h = c_void_p()
        check(self.lib.exp_amd_orient_create(ctx.h, int(number_to_keep), int(target),
                                             int(orient_flags), int(control_flags), float(dT),
                                             float(damping), byref(h)), ctx.h)
        self.h = h
        ctx._children.append(self)

    def set_center(self, x: float, y: float, z: float) -> None:
        v = (c_double * 3)(x, y, z)
        check(self.lib.exp_amd_orient_set_center(self.h, v), self.ctx.h)

    def set_cenvel(self, u: float, v: float, w: float) -> None:
        a = (c_double * 3)(u, v, w)
        check(self.lib.exp_amd_orient_set_cenvel(self.h, a), self.ctx.h)

    def set_linear(self) -> None:
        check(self.lib.exp_amd_orient_set_linear(self.h), self.ctx.h)

    def set_naccel(self, naccel: int) -> None:
        """The ``Naccel`` constructor argument: length of the pseudo-acceleration queue."""
        check(self.lib.exp_amd_orient_set_naccel(self.h, int(naccel)), self.ctx.h)

    def currentAccel(self):
        """``Orient::currentAccel`` -> (accel, omega, domdt) (include/PseudoAccel.H:45-91)."""
        a, o, d = (c_double * 3)(), (c_double * 3)(), (c_double * 3)()
        check(self.lib.exp_amd_orient_accel(self.h, a, o, d), self.ctx.h)
        return np.array(a[:]), np.array(o[:]), np.array(d[:])

    def accumulate(self, time: float, comp: "Component", dtime: float = 0.0) -> None:
        """``Orient::accumulate(time, c)`` (src/Orient.cc:420-747); ``dtime`` is the reference's
        global time step (drift of the user-specified centre)."""
        check(self.lib.exp_amd_orient_accumulate(self.h, float(time), float(dtime), comp.h),
              self.ctx.h)

    def _get(self):
        c, a, b, o, s = ((c_double * 3)(), (c_double * 3)(), (c_double * 9)(), (c_double * 9)(),
                         (c_double * 15)())
        check(self.lib.exp_amd_orient_get(self.h, c, a, b, o, s), self.ctx.h)
        return (np.array(c[:]), np.array(a[:]), np.array(b[:]).reshape(3, 3),
                np.array(o[:]).reshape(3, 3), np.array(s[:]))

    def currentCenter(self) -> np.ndarray:
        return self._get()[0]

    def currentAxis(self) -> np.ndarray:
        return self._get()[1]

    def transformBody(self) -> np.ndarray:
        return self._get()[2]

    def transformOrig(self) -> np.ndarray:
        return self._get()[3]

    def currentE(self) -> float:
        return float(self._get()[4][0])

    def currentUsed(self) -> int:
        return int(self._get()[4][1])

    def currentAxisVar(self) -> float:
        return float(self._get()[4][2])

    def currentCenterVar(self) -> float:
        return float(self._get()[4][3])

    def currentCenterVarZ(self) -> float:
        return float(self._get()[4][4])

    def state(self) -> dict:
        """The quantities ``Orient::logEntry`` writes (src/Orient.cc:749-783)."""
        c, a, b, o, s = self._get()
        return {"center": c, "axis": a, "body": b, "orig": o, "Ecurr": float(s[0]),
                "used": int(s[1]), "sigA": float(s[2]), "sigC": float(s[3]), "sigCz": float(s[4]),
                "mtot": float(s[5]), "axis1": s[6:9].copy(), "center1": s[9:12].copy(),
                "center0": s[12:15].copy()}

    # -- log file and the restart from it (src/Orient.cc:84-335 constructor, :742-785 logEntry) --
    LOG_COLUMNS = ["Time", "E_curr", "Used", "X-axis(reg)", "Y-axis(reg)", "Z-axis(reg)", "X-axis(cur)",
                   "Y-axis(cur)", "Z-axis(cur)", "X-center(anl)", "Y-center(anl)", "Z-center(anl)",
                   "X-center(reg)", "Y-center(reg)", "Z-center(reg)", "X-center(cur)", "Y-center(cur)",
                   "Z-center(cur)", "X-com(cur)", "Y-com(cur)", "Z-com(cur)", "X-com(dif)", "Y-com(dif)",
                   "Z-com(dif)", "X-accel", "Y-accel", "Z-accel", "Omega_X", "Omega_Y", "Omega_Z",
                   "dOmega/dt_X", "dOmega/dt_Y", "dOmega/dt_Z"]

    def openLog(self, path: str, restart: bool = False, tnow: float = 0.0, dtime: float = 0.0,
                Mstep: int = 1, queue_center1: bool = False) -> int:
        """The ``Logfile`` constructor argument and the constructor's restart block: a missing file
        gets the two header rows; an existing one is moved to ``<path>.bak`` and -- with the global
        ``restart`` -- its rows up to ``tnow + 0.1*dtime/Mstep`` are copied back and rebuild the
        estimator's state (include/exp_amd.h).  Call after ``set_naccel``.  Returns the rows taken."""
        from ctypes import c_longlong
        rows = c_longlong(0)
        flags = (1 if restart else 0) | (2 if queue_center1 else 0)
        check(self.lib.exp_amd_orient_open_log(self.h, str(path).encode(), flags, float(tnow),
                                               float(dtime), int(Mstep), byref(rows)), self.ctx.h)
        self._log = str(path)
        return int(rows.value)

    def logHeader(self, path: str) -> None:
        """A fresh log file at ``path`` (any file already there is dropped first)."""
        import os
        if os.path.exists(path):
            os.remove(path)
        self.openLog(path)

    def logEntry(self, time: float, path: str = None, com=(0.0, 0.0, 0.0), com0=(0.0, 0.0, 0.0)) -> None:
        """``Orient::logEntry(time, c)``: one row of 33 columns -- time, Ecurr, used, axis, axis1,
        centre, centre0, centre1, the component's com and com0, pseudo-acceleration, omega, domega/dt
        (column order of the reference's own writer, whose header labels columns 10-15 the other way)."""
        if path is not None and path != getattr(self, "_log", None):
            raise ValueError("Orient.logEntry: rows go to the file given to openLog / logHeader")
        a, b = (c_double * 3)(*com), (c_double * 3)(*com0)
        check(self.lib.exp_amd_orient_log_entry(self.h, float(time), a, b), self.ctx.h)

    def close(self) -> None:
        if self.h:
            self.lib.exp_amd_orient_destroy(self.h)
            self.h = None
            if self in self.ctx._children:
                self.ctx._children.remove(self)


class _Force:
    """PotAccel-shaped methods shared by every force method (src/PotAccel.H:173-288)."""

    ctx: Context
    h: c_void_p

    def set_multistep_level(self, mlevel: int) -> None:
        check(self.lib.exp_amd_force_set_level(self.h, int(mlevel)), self.ctx.h)

    # -- playback (the `playback` / `coefCompute` keys: src/SphericalBasis.cc:155-213, src/Cylinder.cc:
    #    560-618) -------------------------------------------------------------------------------------
    play_back = False

    def set_playback(self, coefs, dtime: float, coef_compute: bool = False) -> None:
        """Drive the force from a stored coefficient series instead of the particles: ``coefs`` is a
        ``SphCoefs`` / ``CylCoefs`` (exp_amd.coefs) or a path to a native stream or HDF5 file.  The
        checks are the reference's: the basis orders must match, the off-grid tolerance is twice the
        time step."""
        if isinstance(coefs, (str, bytes)):
            path = coefs.decode() if isinstance(coefs, bytes) else coefs
            with open(path, "rb") as f:
                magic = f.read(8)
            coefs = self._pb_open(path, magic.startswith(b"\x89HDF"))
        self._pb_check(coefs.getCoefStruct(coefs.Times()[0]))
        coefs.setDeltaT(2.0 * dtime)
        self.playback, self.play_back, self.play_cnew = coefs, True, bool(coef_compute)
        self.lastPlayTime, self.expcoefP, self.stop_signal = -np.inf, None, 0
        self._pb_started = False

    _pending_playback = None

    def start_playback(self, dtime: float) -> None:
        """The ``playback`` / ``coefCompute`` keys taken by ``from_config``, now that the run's time step is known."""
        if self._pending_playback is None:
            raise ValueError("start_playback: no playback key was given")
        path, cc = self._pending_playback
        self.set_playback(path, dtime, cc)
        self._pending_playback = None

    def _pb_first(self, comp: "Component") -> None:
        pass

    def _pb_before_force(self) -> None:
        pass

    def set_mass_scale(self, adb: float) -> None:
        """``component->Adiabatic()`` as evaluated for the current time: every mass the accumulation and the level-change
        differencing read is multiplied by it (src/SphericalBasis.cc:441, :471, :1161; src/Cylinder.cc:834, :1758)."""
        check(self.lib.exp_amd_force_set_mass_scale(self.h, float(adb)), self.ctx.h)

    def set_self_consistent(self, on: bool) -> None:
        """The ``self_consistent`` key (include/exp_amd.h: exp_amd_force_set_self_consistent)."""
        check(self.lib.exp_amd_force_set_self_consistent(self.h, int(bool(on))), self.ctx.h)

    def set_initializing(self, on: bool) -> None:
        """The global ``initializing`` of begin_run (src/begin.cc:80, :129) for hosts that drive the calls themselves."""
        check(self.lib.exp_amd_force_set_initializing(self.h, int(bool(on))), self.ctx.h)

    @property
    def coefs_frozen(self) -> bool:
        return bool(self.lib.exp_amd_force_coefs_frozen(self.h))

    def determine_coefficients(self, comp: "Component", tnow: Optional[float] = None) -> None:
        """``determine_coefficients`` (src/SphericalBasis.cc:600-608, src/Cylinder.cc:898-906): from the
        particles, or -- in playback -- from the coefficient series at ``tnow`` (one interpolation
        per new time, ``:610-680`` / ``:908-946``), plus the particles when ``coefCompute`` is set.
        A component with an adiabatic turn-on (``Component.set_adiabatic``) needs ``tnow`` too."""
        if comp.adiabatic is not None:
            if tnow is None:
                raise ValueError("determine_coefficients: the component has ton / toff / twid set: pass tnow")
            self.set_mass_scale(comp.Adiabatic(tnow))
        if not self.play_back:
            check(self.lib.exp_amd_force_determine_coefficients(self.h, comp.h), self.ctx.h)
            return
        if tnow is None:
            raise ValueError("playback: determine_coefficients needs the current time")
        if not self._pb_started:
            self._pb_first(comp)
            self._pb_started = True
        if tnow > self.lastPlayTime:
            self.lastPlayTime = tnow
            mat, ok = self.playback.interpolate(tnow)
            if not ok:
                self.stop_signal = 1
            self.expcoefP = self._pb_unpack(mat)
        if self.play_cnew:
            check(self.lib.exp_amd_force_determine_coefficients(self.h, comp.h), self.ctx.h)

    def get_acceleration_and_potential(self, comp: "Component", external: bool = False) -> None:
        """Force pass; in playback the played-back set is swapped in for the evaluation and the
        particle-derived one (``coefCompute``) restored after it (src/SphericalBasis.cc:1676-1678,
        :1752-1754; src/Cylinder.cc:1462-1465, :1533-1536)."""
        if not self.play_back or self.expcoefP is None:
            check(self.lib.exp_amd_force_get_acceleration(self.h, comp.h, int(external)), self.ctx.h)
            return
        keep = self._get_flat() if self.play_cnew else None
        self._set_flat(self.expcoefP)
        self._pb_before_force()
        check(self.lib.exp_amd_force_get_acceleration(self.h, comp.h, int(external)), self.ctx.h)
        if keep is not None:
            self._set_flat(keep)

    def compute_multistep_coefficients(self, mdrft: int) -> None:
        check(self.lib.exp_amd_force_compute_multistep_coefficients(self.h, int(mdrft)), self.ctx.h)

    def multistep_reset(self) -> None:
        check(self.lib.exp_amd_force_multistep_reset(self.h), self.ctx.h)

    def Used(self) -> int:
        u = c_longlong()
        check(self.lib.exp_amd_force_used(self.h, byref(u)), self.ctx.h)
        return int(u.value)

    def _get_flat(self, level=None, last=False) -> np.ndarray:
        n = int(self.lib.exp_amd_force_ncoef(self.h))
        out = np.empty(n)
        if level is None:
            check(self.lib.exp_amd_force_get_coefs(self.h, out.ctypes.data_as(c_void_p), n),
                  self.ctx.h)
        else:
            check(self.lib.exp_amd_force_get_level_coefs(self.h, int(level), int(last),
                                                         out.ctypes.data_as(c_void_p), n),
                  self.ctx.h)
        return out

    def _set_flat(self, coef) -> None:
        c = np.ascontiguousarray(coef, dtype=np.float64).reshape(-1)
        assert c.size == int(self.lib.exp_amd_force_ncoef(self.h))
        check(self.lib.exp_amd_force_set_coefs(self.h, c.ctypes.data_as(c_void_p), c.size),
              self.ctx.h)

    def step_kdk(self, comp: "Component", dt: float) -> None:
        """One multistep=0 KDK step (src/step.cc:271-323), fused."""
        check(self.lib.exp_amd_step_kdk(self.h, comp.h, float(dt)), self.ctx.h)

    def step_kdk_n(self, comp: "Component", dt: float, nsteps: int) -> None:
        """``nsteps`` fused KDK steps; pairs of steady-state steps are replayed from a HIP graph
        (include/exp_amd.h: exp_amd_step_kdk_n).  Same results as ``nsteps`` calls of ``step_kdk``."""
        check(self.lib.exp_amd_step_kdk_n(self.h, comp.h, float(dt), int(nsteps)), self.ctx.h)

    def close(self) -> None:
        if self.h:
            self.lib.exp_amd_force_destroy(self.h)
            self.h = None
            if self in self.ctx._children:
                self.ctx._children.remove(self)


class SphereSL(_Force):
    """``sphereSL`` force method: spherical-harmonic x Sturm-Liouville BFE."""

    def __init__(self, ctx: Context, grid: SLGridSph, scale: float = 1.0,
                 rmin: Optional[float] = None, rmax: Optional[float] = None,
                 NO_L0=False, NO_L1=False, EVEN_L=False, EVEN_M=False, M0_only=False,
                 multistep: int = 0, self_consistent: bool = True, FIX_L0: bool = False):
        self.ctx, self.lib, self.grid = ctx, ctx.lib, grid
        # Sphere::Sphere: rmin/rmax are taken from the SL grid (src/Sphere.cc:65-67)
        self.rmin = grid.rmin if rmin is None else rmin
        self.rmax = grid.rmax if rmax is None else rmax
        self.cfg = SphConfig(grid.lmax, grid.nmax, grid.numr, grid.cmap, grid.rmap, scale,
                             self.rmin, self.rmax, grid.xmin, grid.dxi, int(NO_L0), int(NO_L1),
                             int(EVEN_L), int(EVEN_M), int(M0_only), int(multistep))
        keep = [as_f64(grid.xi), as_f64(grid.p0), as_f64(grid.ev), as_f64(grid.ef)]
        h = c_void_p()
        check(self.lib.exp_amd_sph_create(ctx.h, byref(self.cfg), *[k[1] for k in keep], byref(h)),
              ctx.h)
        self.h = h
        self.lmax, self.nmax = grid.lmax, grid.nmax
        self.nrows = (grid.lmax + 1) ** 2
        self.multistep = multistep
        ctx._children.append(self)
        if not self_consistent:
            self.set_self_consistent(False)
        if FIX_L0:
            self.set_fix_l0(True)

    @classmethod
    def from_config(cls, ctx: Context, grid: SLGridSph, conf: dict, multistep: int = 0, nthrds: int = 1) -> "SphereSL":
        """From the reference's YAML keys (``SphericalBasis::valid_keys``, src/SphericalBasis.cc:30-52): every key is
        honoured or refused, none dropped (exp_amd/config.py).  ``nthrds``: the run's global thread count (``ssfrac``'s
        partition of the level list depends on it)."""
        from .config import sphere_from_config
        return sphere_from_config(cls, ctx, grid, conf, multistep, nthrds)

    def set_noise(self, model_file: Optional[str], noiseN: float = 1.0e-6, seedN: int = 0, scale: Optional[float] = None) -> None:
        """The ``NOISE`` mode (src/SphericalBasis.cc:355, :395, :2108-2210): every force evaluation replaces the coefficient
        set by draws from the noise model of ``noise_model_file`` -- ``compute_rms_coefs`` on the host
        (exp_amd.slgrid.compute_rms_coefs), ``update_noise`` inside the library with the reference's own
        std::mt19937 / std::normal_distribution pair.  ``model_file = None`` switches it off."""
        from .slgrid import compute_rms_coefs
        if model_file is None:
            check(self.lib.exp_amd_sph_set_noise(self.h, None, None, 1.0, 0), self.ctx.h)
            self.noise = None
            return
        meanC, rmsC = compute_rms_coefs(self.grid, model_file, float(self.cfg.scale) if scale is None else float(scale))
        km, kr = as_f64(meanC), as_f64(rmsC)
        check(self.lib.exp_amd_sph_set_noise(self.h, km[1], kr[1], float(noiseN), int(seedN) & 0xffffffff), self.ctx.h)
        self.noise = (meanC, rmsC, float(noiseN), int(seedN) & 0xffffffff)

    def set_subset(self, ssfrac: float, nthrds: int = 1) -> None:
        """``ssfrac`` (src/SphericalBasis.cc:149-152, :437-473): with 0 < ssfrac < 1 the coefficients come from a sub-sample
        -- thread ``id`` of ``nthrds`` takes [n id / nthrds, floor(ssfrac n (id + 1) / nthrds)) of the level list (here: the
        caller's particle order), every mass divided by ssfrac.  Any other value switches it off.  Single-level only."""
        check(self.lib.exp_amd_sph_set_subset(self.h, float(ssfrac), int(nthrds)), self.ctx.h)

    def set_fix_l0(self, on: bool = True) -> None:
        """``FIX_L0`` (src/SphericalBasis.cc:1689-1694): the l = 0 row is saved at the next force evaluation and copied
        back into the active set at every later one."""
        check(self.lib.exp_amd_sph_set_fix_l0(self.h, int(bool(on))), self.ctx.h)

    def get_coefs(self, level=None, last=False) -> np.ndarray:
        """(L+1)^2 x nmax, reference real-row order (src/SphericalBasis.cc:513-590)."""
        return self._get_flat(level, last).reshape(self.nrows, self.nmax)

    def set_coefs(self, coef) -> None:
        self._set_flat(coef)

    # -- playback hooks (src/SphericalBasis.cc:155-213, :610-680) ----------------------------------
    def _pb_open(self, path: str, h5: bool):
        from .coefs import SphCoefs
        return SphCoefs.readH5Coefs(path) if h5 else SphCoefs.readNativeCoefs(path)

    def _pb_check(self, first) -> None:
        if first.nmax != self.nmax:
            raise RuntimeError(f"SphericalBasis: nmax for playback [{first.nmax}] does not match "
                               f"specification [{self.nmax}]")
        if first.lmax != self.lmax:
            raise RuntimeError(f"SphericalBasis: Lmax for playback [{first.lmax}] does not match "
                               f"specification [{self.lmax}]")

    def _pb_unpack(self, mat) -> np.ndarray:
        """complex (l, m>=0) rows -> the real-row order of expcoef (:640-651)"""
        from .coefs import complex_to_real_rows
        return complex_to_real_rows(mat, self.lmax)

    def dump_coefs(self, out, time: float = 0.0, scale: Optional[float] = None) -> None:
        """``SphericalBasis::dump_coefs(ostream&)`` (src/SphericalBasis.cc:1829-1879): append the
        current coefficient set to a native coefficient stream (binary file object)."""
        from .basis import SphStruct
        from .coefs import real_rows_to_complex, write_native
        c = SphStruct(self.lmax, self.nmax, self.cfg.scale if scale is None else scale, time,
                      real_rows_to_complex(self.get_coefs(), self.lmax), np.zeros(3), np.eye(3))
        write_native(out, c)

    def dump_coefs_h5(self, file: str, time: float = 0.0, name: str = "", config: str = "", center=None,
                      rotation=None, scale: Optional[float] = None) -> None:
        """``SphericalBasis::dump_coefs_h5`` (src/SphericalBasis.cc:1909-1975; called by ``OutCoef``): the current set as
        one snapshot of pyEXP's HDF5 coefficient file -- appended (``ExtendH5Coefs``) when ``file`` exists, else a new
        file with the component's name, the configuration and the default units {length, mass, time: none} beside the
        container's own G."""
        import os
        from .basis import SphStruct
        from .coefs import SphCoefs, real_rows_to_complex
        cur = SphStruct(self.lmax, self.nmax, self.cfg.scale if scale is None else scale, time,
                        real_rows_to_complex(self.get_coefs(), self.lmax),
                        np.zeros(3) if center is None else np.asarray(center, np.float64),
                        np.eye(3) if rotation is None else np.asarray(rotation, np.float64))
        cs = SphCoefs(name)
        cs.add(cur)
        if os.path.exists(file):
            cs.ExtendH5Coefs(file)
        else:
            cs.setUnits([("length", "none", 1.0), ("mass", "none", 1.0), ("time", "none", 1.0)])
            cs.WriteH5Coefs(file, config=config)

    # -- coefficient covariance by sub-sampling (pyEXP pcavar; expui/BiorthBasis.cc:583-665) ---------
    def cov_enable(self, sampT: int) -> None:
        check(self.lib.exp_amd_sph_cov_enable(self.h, int(sampT)), self.ctx.h)
        self._cov_T = int(sampT)

    def cov_reset(self) -> None:
        check(self.lib.exp_amd_sph_cov_reset(self.h), self.ctx.h)

    def cov_accumulate(self, comp: "Component", used_before: int = 0) -> int:
        """File the particles of ``comp`` (caller order) under their sub-samples; returns the number
        inside the expansion window."""
        acc = c_longlong()
        check(self.lib.exp_amd_sph_cov_accumulate(self.h, comp.h, int(used_before), byref(acc)), self.ctx.h)
        return int(acc.value)

    def cov_get(self) -> dict:
        T, ltot = self._cov_T, (self.lmax + 1) * (self.lmax + 2) // 2
        counts = np.zeros(T, dtype=np.int64)
        masses, mean, covr = np.zeros(T), np.zeros((T, ltot, self.nmax, 2)), np.zeros((T, ltot, self.nmax, self.nmax))
        check(self.lib.exp_amd_sph_cov_get(self.h, counts.ctypes.data, masses.ctypes.data, mean.ctypes.data,
                                           covr.ctypes.data), self.ctx.h)
        return {"counts": counts, "masses": masses, "mean": mean[..., 0] + 1j * mean[..., 1], "covr": covr}

    def _need_density(self) -> None:
        if not getattr(self, "_have_density", False):
            d0 = as_f64(self.grid.d0)
            check(self.lib.exp_amd_sph_set_density(self.h, d0[1]), self.ctx.h)
            self._have_density = True

    def basis(self, r) -> np.ndarray:
        """``SphericalSL::getBasis``'s tables at the radii ``r`` -> [3, lmax+1, nmax, len(r)] = potential,
        density, radial force of every (l, n) (expui/BiorthBasis.cc:960-993)."""
        r = np.ascontiguousarray(np.atleast_1d(r), dtype=np.float64)
        self._need_density()
        out = np.empty((3, self.lmax + 1, self.nmax, r.size))
        check(self.lib.exp_amd_sph_basis(self.h, r.size, r.ctypes.data, out.ctypes.data), self.ctx.h)
        return out

    def window_mass(self, comp: "Component") -> float:
        """Mass of the particles of ``comp`` with rmin <= r <= rmax (``totalMass`` of
        Spherical::accumulate, expui/BiorthBasis.cc:596-607)."""
        m = c_double()
        check(self.lib.exp_amd_sph_window_mass(self.h, comp.h, byref(m)), self.ctx.h)
        return m.value

    FIELD_COORDS = {"spherical": 0, "cylindrical": 1, "cartesian": 2}

    def fields(self, c1, c2, c3, coord: str = "cartesian") -> np.ndarray:
        """Fields of the current coefficient set at points -> [n, 9] = (dens m=0, dens m>0, dens,
        potl m=0, potl m>0, potl, force x3 in the input coordinates): ``Spherical::sph_eval``
        / ``cyl_eval`` / ``crt_eval`` (expui/BiorthBasis.cc:711-816, :930-958)."""
        a, b, c = [np.ascontiguousarray(np.atleast_1d(v), dtype=np.float64) for v in (c1, c2, c3)]
        if not (a.shape == b.shape == c.shape and a.ndim == 1):
            raise ValueError("fields: three 1-d arrays of equal length expected")
        if not getattr(self, "_have_density", False):
            d0 = as_f64(self.grid.d0)
            check(self.lib.exp_amd_sph_set_density(self.h, d0[1]), self.ctx.h)
            self._have_density = True
        out = np.empty((a.size, 9))
        check(self.lib.exp_amd_sph_fields(self.h, a.size, a.ctypes.data, b.ctypes.data,
                                          c.ctypes.data, self.FIELD_COORDS[coord],
                                          out.ctypes.data), self.ctx.h)
        return out


class Cylinder(_Force):
    """``cylinder`` force method: EmpCylSL empirical orthogonal functions (src/Cylinder.cc)."""

    def __init__(self, ctx: Context, grid: EmpCylGrid, rcylmax: Optional[float] = None,
                 EVEN_M: bool = False, multistep: int = 0, self_consistent: bool = True, mlim: int = -1):
        self.ctx, self.lib, self.grid = ctx, ctx.lib, grid
        rcylmax = grid.rmax if rcylmax is None else rcylmax
        self.cfg = CylConfig(grid.mmax, grid.norder, grid.numx, grid.numy, grid.cmapr, grid.cmapz,
                             grid.ascale, grid.hscale, grid.rtable, grid.xmin, grid.dx, grid.ymin,
                             grid.dy, rcylmax, int(EVEN_M), int(multistep))
        tab, ptr = as_f64(grid.tab)
        h = c_void_p()
        check(self.lib.exp_amd_cyl_create(ctx.h, byref(self.cfg), ptr, byref(h)), ctx.h)
        self.h = h
        self.mmax, self.nmax, self.multistep = grid.mmax, grid.norder, multistep
        ctx._children.append(self)
        if not self_consistent:
            self.set_self_consistent(False)
        self.mlim = -1
        if mlim is not None and mlim >= 0:             # `if (mlim>=0) ortho->set_mlim(mlim)` (src/Cylinder.cc:225)
            self.set_mlim(mlim)

    @classmethod
    def from_config(cls, ctx: Context, conf: dict, multistep: int = 0, grid: Optional[EmpCylGrid] = None,
                    condition_on=None) -> "Cylinder":
        """From the reference's YAML keys (``Cylinder::valid_keys``, src/Cylinder.cc:24-80): every key is honoured or
        refused, none dropped (exp_amd/config.py).  ``grid`` None: the EOF tables are built from the keys -- conditioned on
        the analytic disk, or with ``precond: false`` on the particles of ``condition_on`` (a ``Component``, or a (mass, pos)
        pair in the basis' frame): ``Cylinder::determine_coefficients_eof``, src/Cylinder.cc:1202-1249."""
        from .config import cylinder_from_config
        return cylinder_from_config(cls, ctx, conf, multistep, grid, condition_on)

    def set_mlim(self, mlim: int) -> None:
        """The ``mlim`` key (src/Cylinder.cc:225 -> EmpCylSL::set_mlim): harmonics m > mlim take no part in accumulation or
        evaluation (exputil/EmpCylSL.cc:5602, :5317, :5465); their coefficients read back as zero."""
        check(self.lib.exp_amd_cyl_set_mlim(self.h, int(mlim)), self.ctx.h)
        self.mlim = min(int(mlim), self.mmax)

    def get_coefs(self, level=None, last=False):
        """(accum_cos, accum_sin), each (mmax+1) x nmax (exputil/EmpCylSL.cc:4355-4550)."""
        flat = self._get_flat(level, last).reshape(2, self.mmax + 1, self.nmax)
        return flat[0], flat[1]

    def set_coefs(self, cos, sin) -> None:
        self._set_flat(np.stack([np.asarray(cos, dtype=np.float64),
                                 np.asarray(sin, dtype=np.float64)]))

    # -- playback hooks (src/Cylinder.cc:560-618, :908-946, :1825-1860) ----------------------------
    def _pb_open(self, path: str, h5: bool):
        from .coefs import CylCoefs
        return CylCoefs.readH5Coefs(path) if h5 else CylCoefs.readNativeCoefs(path)

    def _pb_check(self, first) -> None:
        if first.nmax != self.nmax:
            raise RuntimeError(f"Cylinder: nmax for playback [{first.nmax}] does not match "
                               f"specification [{self.nmax}]")
        if first.mmax != self.mmax:
            raise RuntimeError(f"Cylinder: mmax for playback [{first.mmax}] does not match "
                               f"specification [{self.mmax}]")

    def _pb_unpack(self, mat) -> np.ndarray:
        return np.stack([np.real(mat), np.imag(mat)])

    def _pb_first(self, comp: "Component") -> None:
        """``Cylinder::compute_grid_mass`` (:1825-1860): mass and count inside rcylmax, once, from the
        particles -- here as the by-product of one accumulation pass."""
        check(self.lib.exp_amd_force_determine_coefficients(self.h, comp.h), self.ctx.h)
        self._pb_cylmass = self.cylmass

    def _pb_before_force(self) -> None:
        self.cylmass = self._pb_cylmass

    @property
    def cylmass(self) -> float:
        m = c_double()
        check(self.lib.exp_amd_cyl_get_cylmass(self.h, byref(m)), self.ctx.h)
        return m.value

    @cylmass.setter
    def cylmass(self, mass: float) -> None:
        check(self.lib.exp_amd_cyl_set_cylmass(self.h, float(mass)), self.ctx.h)

    # -- sub-sample covariance (pyEXP; the `covar` branch of EmpCylSL::accumulate) --------------------
    def cov_enable(self, sampT: int) -> None:
        check(self.lib.exp_amd_cyl_cov_enable(self.h, int(sampT)), self.ctx.h)
        self._cov_T = int(sampT)

    def cov_reset(self) -> None:
        check(self.lib.exp_amd_cyl_cov_reset(self.h), self.ctx.h)

    def cov_accumulate(self, comp: "Component", seq=None) -> int:
        """File the on-grid particles of ``comp`` under sub-sample ``seq % sampT`` (``seq``: the running
        index EmpCylSL::accumulate is handed, caller order; default 0 .. n-1)."""
        acc = c_longlong()
        ptr = None
        if seq is not None:
            sq = np.ascontiguousarray(seq, dtype=np.uint32)
            assert sq.size == comp.n
            ptr = sq.ctypes.data_as(c_void_p)
        check(self.lib.exp_amd_cyl_cov_accumulate(self.h, comp.h, ptr, byref(acc)), self.ctx.h)
        return int(acc.value)

    def cov_get(self) -> dict:
        T, M1, N = self._cov_T, self.mmax + 1, self.nmax
        counts = np.zeros(T, dtype=np.int64)
        masses, vc, mv = np.zeros(T), np.zeros((T, M1, N, 2)), np.zeros((T, M1, N, N, 2))
        check(self.lib.exp_amd_cyl_cov_get(self.h, counts.ctypes.data, masses.ctypes.data, vc.ctypes.data,
                                           mv.ctypes.data), self.ctx.h)
        return {"counts": counts, "masses": masses, "mean": vc[..., 0] + 1j * vc[..., 1],
                "covr": mv[..., 0] + 1j * mv[..., 1]}

    def basis(self, R, z) -> np.ndarray:
        """``Cylindrical::getBasis``'s tables at the points (R, z) -> [4, mmax+1, nmax, npts] = potential,
        density, radial force, vertical force of every (m, n) at phi = 0 (``EmpCylSL::get_all``)."""
        R = np.ascontiguousarray(np.atleast_1d(R), dtype=np.float64)
        z = np.ascontiguousarray(np.atleast_1d(z), dtype=np.float64)
        assert R.shape == z.shape
        out = np.empty((4, self.mmax + 1, self.nmax, R.size))
        check(self.lib.exp_amd_cyl_basis(self.h, R.size, R.ctypes.data, z.ctypes.data, out.ctypes.data), self.ctx.h)
        return out

    def orthocheck(self) -> np.ndarray:
        """``EmpCylSL::orthoCheck`` (exputil/EmpCylSL.cc:7199-7260) -> [mmax+1, nmax, nmax]."""
        out = np.empty((self.mmax + 1, self.nmax, self.nmax))
        check(self.lib.exp_amd_cyl_orthocheck(self.h, out.ctypes.data), self.ctx.h)
        return out

    def dump_coefs_binary(self, out, time: float = 0.0) -> None:
        """``EmpCylSL::dump_coefs_binary`` (exputil/EmpCylSL.cc:5868-5920): append the current
        coefficient set to a native coefficient stream (binary file object)."""
        from .basis import CylStruct
        from .coefs import write_native_cyl
        cc, ss = self.get_coefs()
        write_native_cyl(out, CylStruct(self.grid.mmax, self.grid.norder, time, cc + 1j * ss,
                                        np.zeros(3), np.eye(3)))

    def dump_coefs_h5(self, file: str, time: float = 0.0, name: str = "", config: str = "", center=None,
                      rotation=None) -> None:
        """``Cylinder::dump_coefs_h5`` (src/Cylinder.cc:1625-1690): as the sphere's -- extend an existing file, else
        create it with the name, the configuration and the default units."""
        import os
        from .basis import CylStruct
        from .coefs import CylCoefs
        cc, ss = self.get_coefs()
        cf = cc + 1j * ss
        cf[0] = cc[0]                                    # (m = 0 has no sine part)
        cur = CylStruct(self.grid.mmax, self.grid.norder, time, cf,
                        np.zeros(3) if center is None else np.asarray(center, np.float64),
                        np.eye(3) if rotation is None else np.asarray(rotation, np.float64))
        cs = CylCoefs(name)
        cs.add(cur)
        if os.path.exists(file):
            cs.ExtendH5Coefs(file)
        else:
            cs.setUnits([("length", "none", 1.0), ("mass", "none", 1.0), ("time", "none", 1.0)])
            cs.WriteH5Coefs(file, config=config)

    FIELD_COORDS = {"spherical": 0, "cylindrical": 1, "cartesian": 2}

    def fields(self, c1, c2, c3, coord: str = "cartesian") -> np.ndarray:
        """Fields of the current coefficient set at points -> [n, 9] (dens m=0, dens m>0, dens, potl
        m=0, potl m>0, potl, force x3 in the input coordinates): ``Cylindrical::sph_eval`` /
        ``cyl_eval`` / ``crt_eval`` (expui/BiorthBasis.cc:1749-1849)."""
        a, b, c = [np.ascontiguousarray(np.atleast_1d(v), dtype=np.float64) for v in (c1, c2, c3)]
        if not (a.shape == b.shape == c.shape and a.ndim == 1):
            raise ValueError("fields: three 1-d arrays of equal length expected")
        if not getattr(self, "_have_density", False):
            if getattr(self.grid, "dens", None) is None:
                raise RuntimeError("Cylinder.fields: the EmpCylSL grid has no density tables "
                                   "(rebuild it with exp_amd.empcyl.build_empcyl)")
            d = as_f64(self.grid.dens)
            check(self.lib.exp_amd_cyl_set_density(self.h, d[1]), self.ctx.h)
            self._have_density = True
        out = np.empty((a.size, 9))
        check(self.lib.exp_amd_cyl_fields(self.h, a.size, a.ctypes.data, b.ctypes.data,
                                          c.ctypes.data, self.FIELD_COORDS[coord],
                                          out.ctypes.data), self.ctx.h)
        return out


def do_step_single(force: _Force, comp: Component, dt: float, tnow: Optional[float] = None) -> None:
    """Unfused multistep=0 step, call for call as ``do_step`` (src/step.cc:271-323).  ``tnow`` is
    the reference's global of that name while the step runs -- the END-of-step time (``tnow +=
    dtime`` comes first, ``:274``) -- and is what a force in playback mode is evaluated at."""
    comp.incr_velocity(0.5 * dt)
    comp.incr_position(dt)
    force.set_multistep_level(0)
    if getattr(force, "play_back", False) or comp.adiabatic is not None:
        force.determine_coefficients(comp, tnow)
    else:
        force.determine_coefficients(comp)
    comp.zero_acceleration(0)
    force.get_acceleration_and_potential(comp)
    comp.incr_velocity(0.5 * dt)


class Simulation:
    """The step loop (``do_step`` src/step.cc:67-325, ``begin_run`` src/begin.cc:80-129) over
    components, their force methods and pairwise interactions -- run by the C++ host code of
    libexp_amd.so (``exp_amd_sim_*``)."""

    def __init__(self, ctx: Context, dtime: float, multistep: int = 0, dynfrac=None,
                 shiftlevl: int = 0):
        self.ctx, self.lib = ctx, ctx.lib
        dyn = None
        if dynfrac is not None:
            dyn = (c_double * 5)(*[float(v) for v in dynfrac])
        h = c_void_p()
        check(self.lib.exp_amd_sim_create(ctx.h, int(multistep), float(dtime), dyn, int(shiftlevl),
                                          byref(h)), ctx.h)
        self.h = h
        self._keep = []

    def add_component(self, comp: Component, force: _Force) -> int:
        idx = c_int()
        check(self.lib.exp_amd_sim_add_component(self.h, comp.h, force.h, byref(idx)), self.ctx.h)
        self._keep.append((comp, force))
        return idx.value

    def set_adiabatic(self, index: int, ton: float = -1.0e20, toff: float = 1.0e20, twid: float = 0.1) -> None:
        """The ton / toff / twid keys of component ``index``: the driver multiplies the masses its force method reads by
        Component::Adiabatic() at the driver's time (include/exp_amd.h: exp_amd_sim_set_adiabatic)."""
        check(self.lib.exp_amd_sim_set_adiabatic(self.h, int(index), float(ton), float(toff), float(twid)), self.ctx.h)

    def set_time(self, tnow: float) -> None:
        check(self.lib.exp_amd_sim_set_time(self.h, float(tnow)), self.ctx.h)

    def set_orient(self, index: int, orient: "Orient", dryrun: bool = False, centerlevl: int = -1) -> None:
        """The EJ keys of a component (src/Component.cc:1323-1370): the expansion centre follows the
        estimator, which is fed at every force evaluation of level ``centerlevl``."""
        check(self.lib.exp_amd_sim_set_orient(self.h, int(index), orient.h if orient else None,
                                              int(dryrun), int(centerlevl)), self.ctx.h)

    def set_restart(self, on: bool = True) -> None:
        """The global ``restart``: the estimators take in the first force evaluation's state too."""
        check(self.lib.exp_amd_sim_set_restart(self.h, int(bool(on))), self.ctx.h)

    def set_center_from(self, index: int, source: int) -> None:
        """The component key ``ctr_name`` (``Component::c0``, src/Component.cc:284-310, :3584-3587): component ``index`` takes the
        expansion centre of component ``source`` at every centre update (``source`` < 0: off)."""
        check(self.lib.exp_amd_sim_set_center_from(self.h, int(index), int(source)), self.ctx.h)

    def set_eqmotion(self, on: bool = True) -> None:
        """The global ``eqmotion`` (src/global.cc:54): ``False`` = ``incr_position`` / ``incr_velocity`` return at once
        (src/incpos.cc:75, src/incvel.cc:93): the steps evaluate fields and levels as the time goes on and move nothing."""
        check(self.lib.exp_amd_sim_set_eqmotion(self.h, int(bool(on))), self.ctx.h)

    def add_interaction(self, source: int, target: int) -> None:
        check(self.lib.exp_amd_sim_add_interaction(self.h, int(source), int(target)), self.ctx.h)

    def init(self) -> None:
        check(self.lib.exp_amd_sim_init(self.h), self.ctx.h)

    def step(self, nsteps: int = 1) -> None:
        check(self.lib.exp_amd_sim_step(self.h, int(nsteps)), self.ctx.h)

    @property
    def time(self) -> float:
        return float(self.lib.exp_amd_sim_time(self.h))

    @property
    def last_switches(self) -> int:
        return int(self.lib.exp_amd_sim_last_switches(self.h))

    @property
    def step_switches(self) -> int:
        """level changes summed over the sub-steps of the last ``step`` call"""
        return int(self.lib.exp_amd_sim_step_switches(self.h))

    def close(self) -> None:
        if self.h:
            self.lib.exp_amd_sim_destroy(self.h)
            self.h = None
