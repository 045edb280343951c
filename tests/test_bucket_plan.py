"""CPU test of the gradient-exchange bucket plan at the REAL student's sizes (round-5 review, item 7): the overlapped
ParameterServer exchange (run_distillation.m:88 'tmove', 179-181) pushes the FC filters first -- fc6's bucket is complete as
soon as the backward pass has passed fc6, 82 % of the bytes hide behind conv5 ... conv1's backward -- then the convolution
filters, then the rest; every element exactly once."""
import numpy as np

import cpu_standin


def test_bucket_order_at_the_real_student_sizes():
    undo = cpu_standin.install()
    try:
        from mcncrossmodalemotions_amd import train, zoo
        net = zoo.emoVoxZoo("emovoxceleb-student", scratch=1, lossType="hot-cross-ent", numSeconds=3, numOutputs=8)
        net.pack_params()
        gb = train.GradBuckets(net)
        total = int(net._flat.der.numel())
        assert abs(4 * total - 66.6e6) < 0.4e6, "the student's flat derivative buffer is 66.6 MB (SURVEY 8e)"
        ranges = gb.ranges()
        assert len(ranges) >= 3, ranges
        # canonical order: the order the backward pass reaches the trigger layers
        a, b, trig = gb.buckets[0]
        assert trig == "fc6", gb.buckets
        names = {p._flat_off: k for k, p in net.params.items() if hasattr(p, "_flat_off")}
        inside = [names[o] for o in sorted(names) if a <= o < b]
        assert inside == ["fc6f", "fc7f", "fc8f"], inside
        assert 4 * (b - a) > 0.8 * 4 * total > 50e6, "fc6's bucket carries 82 % of the bytes"
        order = {l.name: i for i, l in enumerate(net.layers)}
        trig_order = [order[t] for _, _, t in gb.buckets]
        assert trig_order == sorted(trig_order, reverse=True), "buckets are pushed in the order the backward pass reaches them"
        # every element exactly once
        cover = np.zeros(total, np.int8)
        for lo, hi in ranges:
            cover[lo:hi] += 1
        assert (cover == 1).all()
    finally:
        undo()
