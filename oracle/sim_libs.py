"""TEST INFRASTRUCTURE — ctypes bindings for the two CPU simulators used as checkers:

  RefSim    : oracle/_ref/libref_sim.so, the REAL reference FreeCar + Box2D 2.4.1 + geometry code
              (built by `make -C oracle ref` in the build container; the prebuilt .so travels to the GPU box)
  OracleSim : oracle/libsim_oracle.so, this repo's C restatement (oracle/sim_oracle.c), contact-free tier

Both expose the same Python interface so tests can run them side by side.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(HERE, "_ref", "libref_sim.so")
ORA_SO = os.path.join(HERE, "libsim_oracle.so")

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build_oracle():
    subprocess.check_call(["make", "-C", HERE, "oracle"], stdout=subprocess.DEVNULL)


def build_ref():
    subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


def _bind(lib, prefix):
    g = lambda n: getattr(lib, prefix + n)
    g("create").restype = C.c_void_p
    g("create").argtypes = [C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, _f32p]
    g("set_action").argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
    g("set_position").argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float]
    g("set_expert").argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
    g("step").argtypes = [C.c_void_p, C.c_float]
    g("get_state").argtypes = [C.c_void_p, _f32p, _u8p, _u8p]
    g("get_body").argtypes = [C.c_void_p, _f32p]
    g("destroy").argtypes = [C.c_void_p]
    return g


class _Sim:
    _lib = None
    _prefix = None
    _path = None

    @classmethod
    def available(cls):
        return os.path.exists(cls._path)

    @classmethod
    def _load(cls):
        if cls._lib is None:
            cls._lib = C.CDLL(cls._path)
            cls._g = staticmethod(_bind(cls._lib, cls._prefix))
        return cls.__dict__["_g"].__func__

    def __init__(self, length, width, x, y, heading, speed, segs=None):
        g = self._load()
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        self.n = len(x)
        segs = np.zeros((0, 4), np.float32) if segs is None else f(segs).reshape(-1, 4)
        self._segs = segs if len(segs) else np.zeros((1, 4), np.float32)
        self.h = g("create")(self.n, f(length), f(width), f(x), f(y), f(heading), f(speed), len(segs), self._segs)

    def set_action(self, i, accel, steer):
        self._g("set_action")(self.h, i, float(accel), float(steer))

    def set_position(self, i, x, y):
        self._g("set_position")(self.h, i, x, y)

    def set_expert(self, i, x, y, heading, speed):
        """Vehicle i is expert-controlled in the NEXT step: after the physics step it is put on this logged state
        (nocturne/cpp/src/scenario.cc:276-283)."""
        self._g("set_expert")(self.h, i, x, y, heading, speed)

    def step(self, dt=0.1):
        self._g("step")(self.h, dt)

    def state(self):
        out = np.zeros((self.n, 6), np.float32)
        cv = np.zeros(self.n, np.uint8)
        ce = np.zeros(self.n, np.uint8)
        self._g("get_state")(self.h, out, cv, ce)
        return out, cv, ce

    def body(self):
        out = np.zeros((self.n, 6), np.float32)
        self._g("get_body")(self.h, out)
        return out

    def close(self):
        if self.h:
            self._g("destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RefSim(_Sim):
    _prefix = "refsim_"
    _path = REF_SO


class OracleSim(_Sim):
    _prefix = "orasim_"
    _path = ORA_SO


def ref_geo():
    lib = C.CDLL(REF_SO)
    lib.refgeo_poly_poly.argtypes = [_f32p, C.c_int, _f32p, C.c_int]
    lib.refgeo_poly_seg.argtypes = [_f32p, C.c_int, _f32p]
    return lib


def oracle_geo():
    lib = C.CDLL(ORA_SO)
    lib.orageo_poly_poly.argtypes = [_f32p, C.c_int, _f32p, C.c_int]
    lib.orageo_poly_seg.argtypes = [_f32p, C.c_int, _f32p]
    return lib
