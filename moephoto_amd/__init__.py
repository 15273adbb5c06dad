"""moephoto_amd -- MI355X (gfx950) engine for MoePhoto's tiled super-resolution / denoise hot path.

Importing the package is cheap; the HIP library is loaded on first use (moephoto_amd._lib.lib()) and
its absence is an error, never a silent fallback.
"""
__all__ = ['config', 'imageProcess', 'models', 'runSR', 'runDN', 'MoeNet_lite2', 'weights', 'dist']
