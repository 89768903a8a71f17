"""Drop-in counterparts of the reference's ``correlation_package/correlation.py`` (dead code in the reference's live tree,
SURVEY.md 2 row 5; only ``networks/._bak/IntVOS_fast.py:264`` ever built one) on the HIP kernels of this library.

  Correlation(pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2, corr_multiply=1)      correlation.py:47-61
  CorrelationFunction(pad_size=3, kernel_size=3, max_displacement=20, stride1=1, stride2=2, corr_multiply=1)   :7-45

Same constructor arguments and defaults, same call shape -- ``Correlation(...)(input1, input2)`` and
``CorrelationFunction(...)(input1, input2)`` (the reference's is an old-style instance Function; here the instance is a plain
callable that applies the static ``autograd.CorrelationFn``) -- same output ``[B, (2r+1)^2, outH, outW]`` with
r = max_displacement // stride2 and the shape rule of correlation_cuda.cc:25-34, in the inputs' dtype (float / half / double
forward, float / half backward).  ``corr_multiply`` is accepted and has no effect, as in the reference: it is a parameter of
the two launchers (``corr_type_multiply``, correlation_cuda_kernel.cu:369,476) that no kernel reads.
"""
import torch.nn as nn

from . import ops


class CorrelationFunction:
    def __init__(self, pad_size=3, kernel_size=3, max_displacement=20, stride1=1, stride2=2, corr_multiply=1):
        self.pad_size, self.kernel_size, self.max_displacement = pad_size, kernel_size, max_displacement
        self.stride1, self.stride2, self.corr_multiply = stride1, stride2, corr_multiply

    def __call__(self, input1, input2):
        return ops.correlation_forward(input1, input2, self.pad_size, self.kernel_size, self.max_displacement,
                                       self.stride1, self.stride2)

    forward = __call__


class Correlation(nn.Module):
    def __init__(self, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2, corr_multiply=1):
        super().__init__()
        self.pad_size, self.kernel_size, self.max_displacement = pad_size, kernel_size, max_displacement
        self.stride1, self.stride2, self.corr_multiply = stride1, stride2, corr_multiply

    def forward(self, input1, input2):
        return CorrelationFunction(self.pad_size, self.kernel_size, self.max_displacement, self.stride1, self.stride2,
                                   self.corr_multiply)(input1, input2)
