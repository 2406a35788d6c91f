from . import banana_rv, d100_gauss, dblgauss_rv, mixture_nd  # noqa: F401
