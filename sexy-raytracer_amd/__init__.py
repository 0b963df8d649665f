"""MI355X-native path-tracing hot path of swishersnaaake/sexy-raytracer.

The package directory name contains a hyphen (it is the reference's name), so
import it with importlib:

    srt = importlib.import_module("sexy-raytracer_amd")

Layout:
  csrc/      hand-written HIP kernels + the C-ABI library (libsrt_hip.so)
  host/      C++ host mirror of the reference's hittable/material/texture API
  abi.py     ctypes mirror of include/srt_hip.h, SceneBuilder
  hipdev.py  ctypes binding of libsrt_hip.so (fails loudly when it is missing); srt.device() returns it
  scenes.py  the BASELINE.json config scenes
  gltf.py    glTF reader with gltfLoad's semantics (model.h:301-460)
"""
import importlib as _il

abi = _il.import_module(__name__ + ".abi")
scenes = _il.import_module(__name__ + ".scenes")


def device():
    """Lazy: loads libsrt_hip.so (hipdev.py); raises if the HIP extension is not built.
    (The binding module is not called device.py: importing a submodule of that name would replace
    this function on the package.)"""
    return _il.import_module(__name__ + ".hipdev")
