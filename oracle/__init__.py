"""CPU oracle for the WSI per-patch inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / timed CPU baseline.  The product path
(``wsi_segmentation_pipeline_amd`` and the drop-in modules at the repo root) never imports
this package and fails loudly when the HIP library is missing.

Parity status: PINNED for the first-party model (``resnets_shift.ResNet`` bag forward,
``models.models.Classifier/Regressor``) and ``contour_ordering.esp`` by golden vectors
generated in-container from the reference itself (``oracle/gen_golden.py`` ->
``tests/golden/*.npz``).  The sliding-window grid / stitch / threshold / bag-builder
restatements follow reference code that cannot be imported here (it needs openslide, cv2,
skimage, mahotas, torchvision - absent from the image); the reference holds no tests or
golden vectors of its own, so those pieces are "parity unpinned" by
the reference and pinned only by hand-derived counts and property tests.
"""
