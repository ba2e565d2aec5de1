"""Inputs for the pinning recipe (README.md): the synthetic models as cv::FileStorage XML, raw 8-bit images, raw 16-bit / float / double images, DT maps."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from partsbaseddetector_amd.model import make_face_like_model, make_image, make_tree_model, make_wide_image  # noqa: E402


def main(out):
    os.makedirs(out, exist_ok=True)
    lines = []
    for tag, model, (w, h, cn), seed in (("tree", make_tree_model([-1, 0, 1, 1, 0], 3, seed=5), (200, 150, 3), 0),
                                         ("gray", make_tree_model([-1, 0, 0], 2, seed=6), (161, 131, 1), 1),
                                         ("face", make_face_like_model(seed=8, ncomp=4, nfilters=30, part_counts=(7, 12)), (160, 120, 3), 2)):
        model.thresh = 0.0     # every root location above 0 becomes a candidate: the pins carry the full rootv maps anyway
        model.save_filestorage(os.path.join(out, f"model_{tag}.xml"))
        make_image(seed, w, h, cn).tofile(os.path.join(out, f"image_{tag}_f32.raw"))
        make_image(seed, w, h, cn).tofile(os.path.join(out, f"image_{tag}_f64.raw"))
        lines.append(f"frame {tag} model_{tag}.xml {w} {h} {cn}")
    # images of the other depths HOGFeatures<T>::pyramid accepts (src/HOGFeatures.cpp:136-146): pins cv::resize / cv::pyrDown of those depths
    # and features<uint16_t | float | double> ("wide <tag> <w> <h> <cn> <cv depth code>")
    for tag, kind, depth, (w, h, cn), seed in (("w16u", np.uint16, 2, (150, 110, 3), 10), ("w32f", np.float32, 5, (97, 131, 1), 11),
                                               ("w64f", np.float64, 6, (203, 77, 3), 12)):
        make_wide_image(kind, seed, w, h, cn).tofile(os.path.join(out, f"image_{tag}.raw"))
        lines.append(f"wide {tag} {w} {h} {cn} {depth}")
    rng = np.random.default_rng(20260927)
    shapes = [(7, 9), (23, 31), (40, 57), (118, 158), (1, 7), (9, 1)]
    for i, (r, c) in enumerate(shapes):
        a = rng.normal(0, 1.5, (r, c)).astype(np.float32)
        if i == 2:
            a = np.round(a)            # exact ties
        with open(os.path.join(out, f"dt_{i}.bin"), "wb") as f:
            np.asarray([r, c, i % 5 - 2, 2 - i % 5], np.int32).tofile(f)
            np.asarray([-0.01 - 0.01 * i, 0.002 * i, -0.02, -0.001 * i], np.float64).tofile(f)
            a.tofile(f)
    lines.append(f"dt {len(shapes)}")
    open(os.path.join(out, "manifest.txt"), "w").write("\n".join(lines) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/pins")
