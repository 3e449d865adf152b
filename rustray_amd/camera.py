"""Host-side mirror of rustray's `Camera` (reference src/camera.rs).

The camera stays on the host: the trace loop only consumes width, height and
the two inverse matrices (reference src/raytracing.rs:282-283, :349, :355,
:369-393), which `Camera.c_struct()` packs into `rr_camera`.
"""
from __future__ import annotations

import math

import numpy as np

from .flat import rr_camera

DEFAULT_FOV_DEG = 90.0            # src/camera.rs:12
DEFAULT_CLIPPING_NEAR = 0.001     # src/camera.rs:14
DEFAULT_CLIPPING_FAR = 1000.0     # src/camera.rs:15
OBLIQUE_CAM_POS = (-0.5, 0.5, 1.0)  # src/camera.rs:10


def approx_equal(a: float, b: float) -> bool:
    """helper::approx_equal (reference src/helper.rs:11-20): compare 6 truncated decimals in f32."""
    f = np.float32(1000000.0)
    return bool(np.trunc(np.float32(a) * f) == np.trunc(np.float32(b) * f))


class Camera:
    def __init__(self):
        # Camera::new, src/camera.rs:44-67
        self.width = 0
        self.height = 0
        self.aspect_ratio = 0.0
        self.fov = float(np.float32(math.radians(DEFAULT_FOV_DEG)))
        self.eye_pos = np.zeros(3)
        self.up = np.array([0.0, 1.0, 0.0])
        self.dir = np.array([0.0, 0.0, -1.0])
        self.clipping_near = DEFAULT_CLIPPING_NEAR
        self.clipping_far = DEFAULT_CLIPPING_FAR
        self.projection = np.eye(4)
        self.view = np.eye(4)
        self.projection_inverse = np.eye(4)
        self.view_inverse = np.eye(4)

    def init(self, width: int, height: int) -> None:
        # src/camera.rs:69-77
        self.width, self.height = int(width), int(height)
        self.aspect_ratio = float(np.float32(width) / np.float32(height))
        self.init_matrices()

    def init_matrices(self) -> None:
        # src/camera.rs:79-90: Perspective3::new + Isometry3::look_at_rh and their inverses
        # the reference's fields are f32 (src/camera.rs:18-41): round what the caller gave before using it; the matrices
        # themselves are evaluated in double and rounded once in c_struct() (include/rustray_host.hpp does the same)
        f32 = lambda v: float(np.float32(v))  # noqa: E731
        a, fovy, zn, zf = f32(self.aspect_ratio), f32(self.fov), f32(self.clipping_near), f32(self.clipping_far)
        t = math.tan(fovy / 2.0)
        p = np.zeros((4, 4))
        p[0, 0] = 1.0 / (a * t)
        p[1, 1] = 1.0 / t
        p[2, 2] = (zf + zn) / (zn - zf)
        p[2, 3] = 2.0 * zf * zn / (zn - zf)
        p[3, 2] = -1.0
        self.projection = p
        pi = np.zeros((4, 4))
        pi[0, 0] = 1.0 / p[0, 0]
        pi[1, 1] = 1.0 / p[1, 1]
        pi[2, 3] = -1.0
        pi[3, 2] = 1.0 / p[2, 3]
        pi[3, 3] = p[2, 2] / p[2, 3]
        self.projection_inverse = pi
        eye = np.asarray(self.eye_pos, dtype=np.float32).astype(np.float64)
        target = eye + np.asarray(self.dir, dtype=np.float32).astype(np.float64)
        z = eye - target
        z = z / np.linalg.norm(z)
        x = np.cross(np.asarray(self.up, dtype=np.float32).astype(np.float64), z)
        x = x / np.linalg.norm(x)
        y = np.cross(z, x)
        v = np.eye(4)
        v[0, :3], v[1, :3], v[2, :3] = x, y, z
        v[0, 3], v[1, 3], v[2, 3] = -x.dot(eye), -y.dot(eye), -z.dot(eye)
        self.view = v
        vi = np.eye(4)
        vi[:3, :3] = v[:3, :3].T
        vi[:3, 3] = eye
        self.view_inverse = vi

    def is_default_cam(self) -> bool:
        # src/camera.rs:92-123
        e, d, u = self.eye_pos, self.dir, self.up
        return (all(approx_equal(e[i], 0.0) for i in range(3))
                and approx_equal(d[0], 0.0) and approx_equal(d[1], 0.0) and approx_equal(d[2], -1.0)
                and approx_equal(u[0], 0.0) and approx_equal(u[1], 1.0) and approx_equal(u[2], 0.0)
                and approx_equal(self.fov, float(np.float32(math.radians(DEFAULT_FOV_DEG))))
                and approx_equal(self.clipping_near, DEFAULT_CLIPPING_NEAR)
                and approx_equal(self.clipping_far, DEFAULT_CLIPPING_FAR))

    def points_in_frustum(self, pts: np.ndarray) -> bool:
        # src/camera.rs:133-140 applied to a batch of points (N,3)
        pv = self.projection @ self.view
        h = np.concatenate([pts, np.ones((len(pts), 1))], axis=1) @ pv.T
        w = h[:, 3]
        return bool(np.all((np.abs(h[:, 0]) <= w) & (np.abs(h[:, 1]) <= w) & (np.abs(h[:, 2]) <= w)))

    def c_struct(self) -> rr_camera:
        c = rr_camera()
        c.width, c.height = self.width, self.height
        c.projection_inverse[:] = np.asarray(self.projection_inverse, np.float32).T.reshape(16).tolist()
        c.view_inverse[:] = np.asarray(self.view_inverse, np.float32).T.reshape(16).tolist()
        return c

    def state(self) -> dict:
        return dict(width=self.width, height=self.height, fov=self.fov, eye_pos=list(map(float, self.eye_pos)),
                    up=list(map(float, self.up)), dir=list(map(float, self.dir)),
                    clipping_near=self.clipping_near, clipping_far=self.clipping_far)

    @classmethod
    def from_state(cls, st: dict) -> "Camera":
        c = cls()
        c.fov = st["fov"]
        c.eye_pos, c.up, c.dir = np.asarray(st["eye_pos"]), np.asarray(st["up"]), np.asarray(st["dir"])
        c.clipping_near, c.clipping_far = st["clipping_near"], st["clipping_far"]
        c.init(st["width"], st["height"])
        return c
