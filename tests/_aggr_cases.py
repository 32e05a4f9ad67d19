"""SoftmaxAggregation / PowerMeanAggregation cases of tests/golden/golden_aggr_v1.pt, shared by the
CPU oracle test and the GPU parity test.  Each entry: (kind, constructor kwargs, use_ptr,
positive inputs)."""
CASES = {
    'softmax_t1': ('softmax', dict(), False, False),
    'softmax_t05_ptr': ('softmax', dict(t=0.5), True, False),
    'softmax_semi': ('softmax', dict(t=2.0, semi_grad=True), False, False),
    'softmax_learn': ('softmax', dict(t=0.7, learn=True), False, False),
    'softmax_learn_channels': ('softmax', dict(t=1.3, learn=True, channels=6), False, False),
    'powermean_p1': ('powermean', dict(), False, False),
    'powermean_p2': ('powermean', dict(p=2.0), False, True),
    'powermean_p3_ptr_clamped': ('powermean', dict(p=3.0), True, False),
    'powermean_learn_channels': ('powermean', dict(p=1.5, learn=True, channels=6), False, True),
}
