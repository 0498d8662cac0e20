"""Scene loading through the C++ host library (reference formats)."""
import ctypes as C

from . import _capi


class LoadedScene:
    """Owns a FlatScene inside libpathed_host.so; `.desc` is a PathedSceneDesc pointer."""

    def __init__(self, scene_path, width, height, asset_root=None):
        self._host = _capi.load_host()
        root = asset_root if asset_root is not None else _capi.REPO_ROOT
        self._handle = self._host.pathed_host_load_scene(scene_path.encode(), width, height, root.encode())
        if not self._handle:
            raise RuntimeError(self._host.pathed_host_last_error().decode())
        self.desc = self._host.pathed_host_scene_desc(self._handle)
        self.width = width
        self.height = height

    def close(self):
        if self._handle:
            self._host.pathed_host_free_scene(self._handle)
            self._handle = None

    def __del__(self):
        self.close()

    @property
    def n_triangles(self):
        return int(self.desc.contents.n_triangles)

    @property
    def n_spheres(self):
        return int(self.desc.contents.n_spheres)

    @property
    def n_materials(self):
        return int(self.desc.contents.n_materials)
