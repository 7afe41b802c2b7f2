"""CPU checks of the parity-test infrastructure itself (tests/parity_tools.py, tests/structured.py): a helper that
silently measures the wrong thing would weaken every GPU assertion built on it."""
import torch

from parity_tools import bf16_result_error
from structured import fit_centroid_head, structured_images


def _bf16(x):
    return torch.tensor(x, dtype=torch.float32).to(torch.bfloat16)


def test_bf16_result_error_counts_ulps_of_the_result_plus_accumulation_noise():
    x = _bf16([0.5, 1.0, 0.5, 0.25, 0.0])
    terms = torch.zeros(5)
    # neighbours in bf16 (one ulp apart, also across a power of two) -> ratio 1; identical -> 0
    a = _bf16([0.5, 1.0078125, 0.498046875, 0.25, 0.0])
    b = _bf16([0.498046875, 1.0, 0.498046875, 0.2490234375, 0.0])
    ratio, worst = bf16_result_error(a, b, x, terms)
    assert 0.999 < ratio <= 1.0 and worst["ratio"] == ratio            # (the 2^-20 |x| allowance shaves 1e-4 off)
    # three ulps apart -> ~3: a real error is not hidden
    a3 = _bf16([0.75])
    b3 = _bf16([0.75 + 3 * 2.0 ** -8])
    r3, _ = bf16_result_error(a3, b3, _bf16([0.75]), torch.zeros(1))
    assert 2.9 < r3 < 3.1
    # a sum that cancels to ~0 (x = 0, delta = 3e-8 formed from terms of ~1e-3): two correct fp32 evaluations may differ by
    # 5e-10, which is many bf16 ulps OF THE RESULT but far inside the accumulation noise of the operands
    ac, bc = _bf16([-3.306e-8]), _bf16([-3.260e-8])
    rc, _ = bf16_result_error(ac, bc, _bf16([0.0]), torch.tensor([2e-3]))
    assert rc < 1.0
    # ... while the same difference on a sum of tiny terms is an error
    rc2, _ = bf16_result_error(ac, bc, _bf16([0.0]), torch.tensor([1e-9]))
    assert rc2 > 1.0


def test_structured_images_are_seeded_and_class_structured():
    a, la = structured_images(24, classes=4, seed=7, size=32)
    b, lb = structured_images(24, classes=4, seed=7, size=32)
    c, _ = structured_images(24, classes=4, seed=7, size=32, draw=1)
    assert torch.equal(a, b) and torch.equal(la, lb) and la.tolist() == [i % 4 for i in range(24)]
    assert a.shape == (24, 3, 32, 32) and float(a.min()) >= 0.0 and float(a.max()) <= 1.0
    assert not torch.equal(a, c)                                  # a held-out draw of the same classes
    same = (a[0] - a[4]).abs().mean()                             # images 0 and 4: same class
    other = (a[0] - a[1]).abs().mean()                            # 0 and 1: different classes
    assert float(c[0:4].mean()) > 0 and float(same) < float(other)
    assert float((a[0] - c[0]).abs().mean()) < float(other)       # the held-out image of class 0 is closer to class 0


def test_fit_centroid_head_classifies_every_image_with_the_requested_margin():
    from dl_attack_on_imagenet_amd import zoo
    images, labels = structured_images(12, classes=3, seed=5, size=64, noise=0.05)
    model = zoo.build_classifier("resnet18", seed=2)
    margins, pred = fit_centroid_head(model, images, labels, 3, "cpu", target_margin=7.0)
    assert pred.tolist() == labels.tolist()
    assert abs(float(margins.median()) - 7.0) < 1e-3 and float(margins.min()) > 0
    with torch.no_grad():
        out = model(images)
    assert out.argmax(1).tolist() == labels.tolist() and float(out[:, 3:].max()) < -1e3      # unused classes never win
