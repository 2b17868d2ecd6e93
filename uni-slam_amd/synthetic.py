"""
A synthetic RGB-D sequence for exercising the tracking / mapping drivers without a dataset (there is none in the build
container): an analytic indoor scene (a box room seen from inside, a sphere and a block standing in it), sphere-traced
depth, a smooth procedural colour field, and a camera moving on a small loop -- the quantities a reader of
src/utils/datasets.py:49-138 hands to the SLAM loop: (idx, color[H,W,3], depth[H,W], c2w[4,4], rays_d[H,W,3]).
Camera convention as in the reference (common.py:35-46): x right, y up, the camera looks along -z.
"""
import math

import torch

from .common import get_camera_rays


class SyntheticRoom:
    # path="loop": a closed circle of radius 0.7 m (turn 1.637 deg per 2 cm step: 220 frames per round) in the free half of the room,
    # at least 0.2 m from every surface -- for runs longer than the default arc allows (see `clearance`)
    LOOP = dict(start=(4.2, 1.3, -0.1), yaw0_deg=-53.13, turn_deg=1.637)

    def __init__(self, n_frames=40, H=120, W=160, fov_deg=80.0, device="cuda:0", room=((0.0, 6.0), (-0.6, 3.0), (-1.2, 1.0)),
                 step_m=0.02, turn_deg=0.8, seed=0, tex_freq=1.0, start=(1.3, 1.2, -0.1), yaw0_deg=0.0, path=None, clearance=0.05):
        """
        clearance: every camera position must keep this distance (m) from every surface, else ValueError.  r5: the default arc (radius
        1.43 m from (1.3, 1.2)) passes 0.14 m from the sphere at frame ~105 and LEAVES THE ROOM through the wall y = 3.0 at frame 194 --
        a 300-frame run on it showed a 'drift' of 8.6 cm from frame 200 on that was the camera looking at the room from outside.
        """
        if path == "loop":
            start, yaw0_deg, turn_deg = self.LOOP["start"], self.LOOP["yaw0_deg"], self.LOOP["turn_deg"]
        elif path is not None:
            raise ValueError(f"SyntheticRoom: unknown path {path!r}")
        self.n_img, self.H, self.W, self.device = n_frames, H, W, torch.device(device)
        self.fx = self.fy = 0.5 * W / math.tan(math.radians(fov_deg) / 2)
        self.cx, self.cy = (W - 1) / 2.0, (H - 1) / 2.0
        self.tex_freq = float(tex_freq)                                          # > 1: finer colour pattern (long runs along bare walls need it)
        self.room = torch.tensor(room, dtype=torch.float32, device=self.device)
        self.sphere_c = torch.tensor([3.6, 0.9, -0.5], device=self.device); self.sphere_r = 0.55
        self.block_c = torch.tensor([2.2, 2.1, -0.7], device=self.device)
        self.block_h = torch.tensor([0.45, 0.35, 0.5], device=self.device)
        self.dirs = get_camera_rays(H, W, self.fx, self.fy, self.cx, self.cy).to(self.device)      # [H,W,3] camera frame
        self.poses = self._trajectory(n_frames, step_m, turn_deg, seed, start, yaw0_deg, level=(path == "loop"))
        if clearance is not None and n_frames > 0:
            d = self.sdf(self.poses[:, :3, 3])
            bad = torch.nonzero(d < clearance)
            if bad.numel():
                k = int(bad[0])
                raise ValueError(f"SyntheticRoom: the camera path comes within {float(d[k]):.3f} m of a surface at frame {k} (clearance {clearance} m; "
                                 f"negative: inside an object or outside the room).  Use fewer frames, path='loop', or another start / turn_deg.")
        self._cache = {}

    # ---- scene ------------------------------------------------------------------------------------------------
    def sdf(self, p):
        lo, hi = self.room[:, 0], self.room[:, 1]
        inside = torch.minimum(p - lo, hi - p).min(-1)[0]                       # distance to the nearest wall (>0 inside)
        sph = (p - self.sphere_c).norm(dim=-1) - self.sphere_r
        q = (p - self.block_c).abs() - self.block_h
        blk = q.clamp(min=0).norm(dim=-1) + q.max(-1)[0].clamp(max=0)
        return torch.minimum(inside, torch.minimum(sph, blk))

    def color(self, p):
        k = torch.tensor([2.1, 1.7, 2.9], device=p.device) * self.tex_freq
        base = 0.5 + 0.5 * torch.sin(p * k + torch.tensor([0.3, 1.1, 2.0], device=p.device))
        tint = 0.5 + 0.5 * torch.sin((p[..., :1] + p[..., 1:2] * 0.7 - p[..., 2:3] * 0.4) * 1.3)
        return (0.75 * base + 0.25 * tint).clamp(0, 1)

    # ---- camera path --------------------------------------------------------------------------------------------
    def _trajectory(self, n, step_m, turn_deg, seed, start=(1.3, 1.2, -0.1), yaw0_deg=0.0, level=False):
        """camera starts at `start` looking along yaw0 (default: near the room centre, along +x) and drifts on a slow arc; z is up in the world"""
        poses = []
        pos = torch.tensor([float(v) for v in start]); yaw = math.radians(yaw0_deg); pitch = -0.05
        for k in range(n):
            cy_, sy_ = math.cos(yaw), math.sin(yaw)
            fwd = torch.tensor([cy_ * math.cos(pitch), sy_ * math.cos(pitch), math.sin(pitch)])
            up0 = torch.tensor([0.0, 0.0, 1.0])
            right = torch.linalg.cross(fwd, up0); right = right / right.norm()
            up = torch.linalg.cross(right, fwd)
            c2w = torch.eye(4)
            c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, up, -fwd, pos
            poses.append(c2w.clone())
            move = step_m * (0.8 * fwd + 0.6 * right)
            if level:                                                           # (a closed loop: the pitch does not carry the camera down)
                move[2] = 0.0
            pos = pos + move + torch.tensor([0.0, 0.0, 0.004 * math.sin(0.3 * k)])
            yaw += math.radians(turn_deg); pitch += math.radians(0.1 * math.cos(0.25 * k))
        return torch.stack(poses).to(self.device)

    # ---- rendering ----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def render(self, c2w, n_steps=96):
        rd = torch.sum(self.dirs[..., None, :] * c2w[:3, :3], -1)                # [H,W,3], not normalised: depth is along -z
        ro = c2w[:3, 3].expand_as(rd)
        scale = rd.norm(dim=-1)
        t = torch.zeros(self.H, self.W, device=self.device)
        for _ in range(n_steps):
            d = self.sdf(ro + rd * t[..., None])
            t = t + d / scale                                                     # sphere tracing in units of the z-depth
        p = ro + rd * t[..., None]
        return self.color(p), t

    def __len__(self):
        return self.n_img

    def __getitem__(self, idx):
        """(idx, color [H,W,3], depth [H,W], c2w [4,4], rays_d [H,W,3]) -- the reference's frame_reader item"""
        if idx not in self._cache:
            self._cache[idx] = self.render(self.poses[idx])
        color, depth = self._cache[idx]
        return idx, color, depth, self.poses[idx], self.dirs
