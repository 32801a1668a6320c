from .ms_deform_attn_func import (BankSlices, MSDeformAttnFunction, MSDeformAttnFusedFunction,  # noqa: F401
                                  ValueBank)
