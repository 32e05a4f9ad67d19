from .linear import HeteroLinear, Linear
