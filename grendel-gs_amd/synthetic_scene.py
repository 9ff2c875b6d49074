"""Seeded synthetic scenes and cameras for tests, smoke() and bench.py.

There is no dataset on the GPU box, so the measured workload is fabricated: Gaussians already
*activated* the way ``GaussianModel.get_*`` hands them to the op (scene/gaussian_model.py:109-129 of
the reference) and cameras in the reference's matrix conventions (scene/cameras.py:84-100,
utils/graphics_utils.py:42-76: world_view_transform = W2C^T, full_proj = wv @ P^T,
camera_center = inverse(wv)[3,:3], znear 0.01, zfar 100).
"""
import math

import torch


class SyntheticCamera:
    """duck-type of scene.cameras.Camera for the fields the hot path reads
    (gaussian_renderer/__init__.py:927-940, workload_division.py:812,880)."""

    def __init__(self, uid, width, height, fx=None, fy=None, R=None, T=None, device="cpu"):
        self.uid = uid
        self.image_name = f"synthetic_{uid:05d}"
        self.image_width = int(width)
        self.image_height = int(height)
        fx = 0.9 * width if fx is None else fx
        fy = fx if fy is None else fy
        self.FoVx = 2.0 * math.atan(width / (2.0 * fx))
        self.FoVy = 2.0 * math.atan(height / (2.0 * fy))
        self.znear, self.zfar = 0.01, 100.0
        R = torch.eye(3, dtype=torch.float64) if R is None else R.to(torch.float64)
        T = torch.zeros(3, dtype=torch.float64) if T is None else T.to(torch.float64)
        # getWorld2View2(R, t): Rt[:3,:3] = R^T, Rt[:3,3] = t  (utils/graphics_utils.py:42-54)
        Rt = torch.zeros(4, 4, dtype=torch.float64)
        Rt[:3, :3] = R.t()
        Rt[:3, 3] = T
        Rt[3, 3] = 1.0
        wv = Rt.to(torch.float32).t().contiguous()
        tanx, tany = math.tan(self.FoVx / 2), math.tan(self.FoVy / 2)
        top, right = tany * self.znear, tanx * self.znear
        P = torch.zeros(4, 4)
        P[0, 0] = 2.0 * self.znear / (2.0 * right)
        P[1, 1] = 2.0 * self.znear / (2.0 * top)
        P[3, 2] = 1.0
        P[2, 2] = self.zfar / (self.zfar - self.znear)
        P[2, 3] = -(self.zfar * self.znear) / (self.zfar - self.znear)
        proj = P.t().contiguous()
        self.world_view_transform = wv.to(device)
        self.projection_matrix = proj.to(device)
        self.full_proj_transform = (wv.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).to(device)
        self.camera_center = wv.inverse()[3, :3].contiguous().to(device)
        self.original_image = None
        self.original_image_backup = None

    def to(self, device):
        for k in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
            setattr(self, k, getattr(self, k).to(device))
        return self


def make_gaussians(n, width, height, seed=0, fx=None, device="cpu", sh_rest_sigma=0.1, scale_coef=0.004,
                   opacity_logit_mean=0.0, opacity_logit_std=2.0):
    """Activated Gaussian attributes in view space of the identity camera (R=I, T=0).

    z ~ U(2,10); x,y ~ U(-1.15,1.15) * z * tanfov (about 13 % outside the frustum, exercising the
    cull); log-scale ~ N(log(scale_coef*z), 0.5^2) per axis; unit quaternions from N(0,1)^4;
    opacity = sigmoid(N(opacity_logit_mean, opacity_logit_std^2)) (N(0, 2^2); the knob exists for the
    low-opacity sensitivity run); SH dc ~ U(-1,1)/0.28209, rest ~ N(0, 0.1^2)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    fx = 0.9 * width if fx is None else fx
    tanx = width / (2.0 * fx)
    tany = height / (2.0 * fx)
    z = torch.rand(n, generator=g) * 8.0 + 2.0
    x = (torch.rand(n, generator=g) * 2.3 - 1.15) * z * tanx
    y = (torch.rand(n, generator=g) * 2.3 - 1.15) * z * tany
    means3D = torch.stack([x, y, z], dim=1)
    log_s = torch.log(scale_coef * z)[:, None] + 0.5 * torch.randn(n, 3, generator=g)
    scales = torch.exp(log_s)
    q = torch.randn(n, 4, generator=g)
    rotations = q / q.norm(dim=1, keepdim=True)
    opacities = torch.sigmoid(opacity_logit_mean + opacity_logit_std * torch.randn(n, 1, generator=g))
    dc = (torch.rand(n, 1, 3, generator=g) * 2.0 - 1.0) / 0.28209479177387814
    rest = sh_rest_sigma * torch.randn(n, 15, 3, generator=g)
    shs = torch.cat([dc, rest], dim=1).contiguous()
    return dict(
        means3D=means3D.contiguous().to(device),
        scales=scales.contiguous().to(device),
        rotations=rotations.contiguous().to(device),
        shs=shs.to(device),
        opacities=opacities.contiguous().to(device),
    )


def orbit_cameras(n_views, width, height, device="cpu", centroid_z=6.0):
    """B cameras rotated about the cloud centroid (0,0,centroid_z) by 360*k/B degrees about the y axis."""
    cams = []
    c = torch.tensor([0.0, 0.0, centroid_z], dtype=torch.float64)
    for k in range(n_views):
        th = 2.0 * math.pi * k / n_views
        # world->camera rotation (math convention) about y
        Rw2c = torch.tensor(
            [[math.cos(th), 0.0, -math.sin(th)], [0.0, 1.0, 0.0], [math.sin(th), 0.0, math.cos(th)]],
            dtype=torch.float64,
        )
        # p_cam = Rw2c (p - c) + c  ->  t = c - Rw2c c ; Camera stores R = Rw2c^T (COLMAP-style transposed R)
        t = c - Rw2c @ c
        cams.append(SyntheticCamera(k, width, height, R=Rw2c.t(), T=t, device=device))
    return cams


def make_gt_image(width, height, seed=1, device="cpu"):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.randint(0, 256, (3, height, width), generator=g, dtype=torch.uint8).to(device)


class SyntheticGaussianModel(torch.nn.Module):
    """duck-type of scene.gaussian_model.GaussianModel for the hot path: raw parameters in the
    reference's layout (scene/gaussian_model.py:219-242) and the activations of its getters
    (scene/gaussian_model.py:109-129).  `rank`/`world_size` keep the contiguous shard this rank owns
    (scene/gaussian_model.py:181-194)."""

    def __init__(self, n_total, width, height, seed=0, sh_degree=3, rank=0, world_size=1, device="cpu",
                 scale_coef=0.004, opacity_logit_mean=0.0, opacity_logit_std=2.0, on_device=False, shards=None):
        super().__init__()
        self.active_sh_degree = sh_degree
        self.max_sh_degree = 3
        if on_device:
            self._init_on_device(n_total, width, height, seed, rank, world_size, device, scale_coef,
                                 opacity_logit_mean, opacity_logit_std, shards)
            return
        g = make_gaussians(n_total, width, height, seed=seed, scale_coef=scale_coef,
                           opacity_logit_mean=opacity_logit_mean, opacity_logit_std=opacity_logit_std)
        chunk = (n_total + world_size - 1) // world_size
        l, r = rank * chunk, min((rank + 1) * chunk, n_total)
        sl = slice(l, r)
        P = torch.nn.Parameter
        op = g["opacities"][sl].clamp(1e-6, 1 - 1e-6)
        self._xyz = P(g["means3D"][sl].clone().to(device))
        self._features_dc = P(g["shs"][sl, 0:1].clone().contiguous().to(device))
        self._features_rest = P(g["shs"][sl, 1:].clone().contiguous().to(device))
        self._scaling = P(torch.log(g["scales"][sl]).to(device))
        self._rotation = P(g["rotations"][sl].clone().to(device))
        self._opacity = P(torch.log(op / (1 - op)).to(device))

    def _init_on_device(self, n_total, width, height, seed, rank, world_size, device, scale_coef, op_mean, op_std,
                        shards):
        """the same distributions generated ON the device, shard by shard (the multi-GPU bench scenes hold up to 40 M
        Gaussians: 9.4 GB of parameters are not drawn on the host and copied).  Shard r of W is a pure function of
        (seed, r, W): a rank builds its own; the single-GPU run of the same workload concatenates shards 0..W-1."""
        chunk = (n_total + world_size - 1) // world_size
        shards = [rank] if shards is None else list(shards)
        parts = {k: [] for k in ("xyz", "dc", "rest", "scaling", "rotation", "opacity")}
        fx = 0.9 * width
        tanx, tany = width / (2.0 * fx), height / (2.0 * fx)
        for r in shards:
            n = min((r + 1) * chunk, n_total) - r * chunk
            g = torch.Generator(device=device)
            g.manual_seed(seed * 100003 + r * 1009 + world_size)

            def U(*shape):
                return torch.rand(*shape, generator=g, device=device)

            def Nrm(*shape):
                return torch.randn(*shape, generator=g, device=device)

            z = U(n) * 8.0 + 2.0
            x = (U(n) * 2.3 - 1.15) * z * tanx
            y = (U(n) * 2.3 - 1.15) * z * tany
            parts["xyz"].append(torch.stack([x, y, z], dim=1))
            parts["scaling"].append(torch.log(scale_coef * z)[:, None] + 0.5 * Nrm(n, 3))
            q = Nrm(n, 4)
            parts["rotation"].append(q / q.norm(dim=1, keepdim=True))
            parts["opacity"].append((op_mean + op_std * Nrm(n, 1)).clamp(-13.8, 13.8))  # logit of [1e-6, 1 - 1e-6]
            parts["dc"].append((U(n, 1, 3) * 2.0 - 1.0) / 0.28209479177387814)
            parts["rest"].append(0.1 * Nrm(n, 15, 3))
        P = torch.nn.Parameter
        cat = {k: (v[0] if len(v) == 1 else torch.cat(v, 0)).contiguous() for k, v in parts.items()}
        self._xyz, self._features_dc, self._features_rest = P(cat["xyz"]), P(cat["dc"]), P(cat["rest"])
        self._scaling, self._rotation, self._opacity = P(cat["scaling"]), P(cat["rotation"]), P(cat["opacity"])

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    def param_groups(self):
        """the reference's Adam groups and learning rates (scene/gaussian_model.py:244-292,
        arguments/__init__.py:109-118)"""
        return [
            {"params": [self._xyz], "lr": 0.00016, "name": "xyz"},
            {"params": [self._features_dc], "lr": 0.0025, "name": "f_dc"},
            {"params": [self._features_rest], "lr": 0.0025 / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": 0.05, "name": "opacity"},
            {"params": [self._scaling], "lr": 0.005, "name": "scaling"},
            {"params": [self._rotation], "lr": 0.001, "name": "rotation"},
        ]


class SyntheticDataset:
    def __init__(self, cameras):
        self.cameras = cameras
