"""mmfusion — MI355X-native host binding for the cross-modal fusion path.

``mmfusion.lib``   ctypes loader of ``libmmfusion.so`` (the C-ABI in ``include/mmfusion.h``)
``mmfusion.ops``   ``torch.autograd.Function`` wrappers that enqueue the HIP kernels
``mmfusion.synth`` seeded synthetic features / parameters
``mmfusion.dp``    data-parallel gradient arena + RCCL all-reduce

Importing the package never touches the GPU; the shared library is loaded on first use and
its absence is a hard error (there is no CPU or eager fallback for the product path).
"""
__version__ = "0.1.0"
