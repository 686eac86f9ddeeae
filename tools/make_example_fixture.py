"""Copies the reference's example depth frame (data, not code) into tests/golden/ as a
compressed uint16 array so that GPU-box tests, which cannot see /root/reference, can use it.
Source: /root/reference/example/1305031459.274941.png (TUM freiburg1_desk, 16UC1 640x480);
pose and thresholds from README.md:51 and src/hope_node.cpp:160-161."""
import numpy as np
from PIL import Image

d = np.array(Image.open("/root/reference/example/1305031459.274941.png"))
assert d.dtype == np.uint16 and d.shape == (480, 640)
np.savez_compressed("tests/golden/tum_example_depth.npz", depth=d,
                    t=np.array([-0.2171, -0.0799, 1.3959], np.float32),
                    q=np.array([-0.8445, -0.0451, 0.0954, 0.5251], np.float32))
print("valid pixels", int(((d.astype(np.float32) / np.float32(5000) > np.float32(0.3)) &
                           (d.astype(np.float32) / np.float32(5000) < 8.0)).sum()))
