from .base import Aggregation
from .basic import (MaxAggregation, MeanAggregation, MinAggregation, MulAggregation,
                    SumAggregation, aggregation_resolver)
from .deeper import PowerMeanAggregation, SoftmaxAggregation
from .fused import FusedAggregation, MultiAggregation, StdAggregation, VarAggregation

__all__ = ['Aggregation', 'SumAggregation', 'MeanAggregation', 'MaxAggregation',
           'MinAggregation', 'MulAggregation', 'VarAggregation', 'StdAggregation',
           'FusedAggregation', 'MultiAggregation', 'SoftmaxAggregation', 'PowerMeanAggregation',
           'aggregation_resolver']
