from .base import Aggregation
from .basic import (MaxAggregation, MeanAggregation, MinAggregation, MulAggregation,
                    SumAggregation, aggregation_resolver)

__all__ = ['Aggregation', 'SumAggregation', 'MeanAggregation', 'MaxAggregation',
           'MinAggregation', 'MulAggregation', 'aggregation_resolver']
