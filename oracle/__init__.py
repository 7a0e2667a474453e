"""CPU oracle for the U-Net anomaly-segmentation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the reported CPU
baseline -- never as the thing measured or shipped.  The product package
(``tiaozhanbei_unet_amd``) must not import this package and fails loudly when
its HIP library is missing.

Parity pin: the reference ships no tests, golden vectors or fixtures for this
path, so the oracle is pinned against OUTPUTS OF THE
REFERENCE ITSELF, produced in the build container by importing
``/root/reference/src/model.py`` and ``src/train_utils.py``
(``tools/make_goldens.py``) and committed as small ``.npz`` fixtures under
``tests/golden/``.  ``tests/test_oracle_golden.py`` checks every oracle function
against those fixtures.

``pil_oracle.py`` (round 4) restates the image transforms of the loaders
(src/dataset.py:130-154, src/kolektorsdd_dataset.py:133-155).  The reference
runs them through torchvision on PIL images, i.e. inside Pillow -- a
third-party dependency, not part of ``/root/reference``; the restatement is
pinned by fixtures Pillow 12.2.0 itself produced in the build container
(``tools/make_goldens_aug.py`` -> ``tests/golden/aug_pil.npz``,
``tests/test_oracle_aug_golden.py``).
"""
