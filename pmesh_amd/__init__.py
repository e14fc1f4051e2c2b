"""pmesh_amd — the particle-mesh cycle of MP-Gadget/pmesh on AMD MI355X (gfx950).

    from pmesh_amd.pm import ParticleMesh, RealField, ComplexField
    from pmesh_amd.window import ResampleWindow, Affine, CIC, TSC, PCS
    from pmesh_amd.domain import GridND, Layout
    from pmesh_amd.transfer import Transfer

Host code is Python; all arithmetic is in libpmesh_amd.so (hand-written HIP
kernels + rocFFT behind the C ABI of include/pmesh_amd.h).  Importing the
package does not touch the GPU; the first operation does, and raises if the
library or the device is missing — there is no CPU fallback.
"""
__version__ = '0.1.0'
