"""Import shim for the `gsplat` names the reference imports unconditionally at
gaussian_renderer/__init__.py:18-25.  gsplat is a third-party ALTERNATIVE backend (selected only by
`--backend gsplat`) whose source is not part of the reference tree; this build provides the default
diff_gaussian_rasterization backend only, so the names exist and raise when called."""


def _unavailable(name):
    def fn(*a, **k):
        raise NotImplementedError(f"gsplat.{name}: the gsplat backend is not part of this build (use the default backend)")

    fn.__name__ = name
    return fn


rasterization = _unavailable("rasterization")
fully_fused_projection = _unavailable("fully_fused_projection")
spherical_harmonics = _unavailable("spherical_harmonics")
isect_tiles = _unavailable("isect_tiles")
isect_offset_encode = _unavailable("isect_offset_encode")
rasterize_to_pixels = _unavailable("rasterize_to_pixels")
