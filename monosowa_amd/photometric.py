"""Photometric distortion of the training images (dataset.aug_pd) -- the reference's SSD-style augmentation
(lib/datasets/kitti/pd.py:114-416: RandomBrightness, RandomContrast, ConvertColor, RandomSaturation, RandomHue,
RandomLightingNoise / SwapChannels, PhotometricDistort), used at lib/datasets/kitti/kitti_dataset.py:94,182-185 and switched
on by the reference's shipped mixed-dataset config (checkpoints/best_kitti_k360_to_kitti/monodetr_kk360_05.yaml:18).

Restated with numpy in the reference's DRAW ORDER on the global ``numpy.random`` stream (pd.py imports ``from numpy import
random``): per image
    randint(2) [+ uniform(-32, 32)]                       brightness: += delta
    randint(2)                                            1: contrast first, 0: contrast last
    [randint(2) [+ uniform(0.5, 1.5)]]                    contrast (first position)
    randint(2) [+ uniform(0.5, 1.5)]                      saturation (HSV channel 1)
    randint(2) [+ uniform(-18, 18)]                       hue (HSV channel 0, wrapped into [0, 360])
    [randint(2) [+ uniform(0.5, 1.5)]]                    contrast (last position)
    randint(2) [+ randint(6)]                             lighting noise: a permutation of the three channels
The one third-party operation is OpenCV's ``cv2.cvtColor(float32 image, COLOR_BGR2HSV / COLOR_HSV2BGR)``.  OpenCV is absent
from this image (no binary to run, no source under /root/reference), so the conversion restates OpenCV's float KERNEL as its
source reads (imgproc color_hsv: ``RGB2HSV_f`` / ``HSV2RGB_f``) rather than the idealised formula of its documentation -- the two
differ where it matters here: S = (V - min) / (|V| + FLT_EPSILON) and H = (..) * 60 / ((V - min) + FLT_EPSILON), i.e. a pixel that
RandomBrightness has driven NEGATIVE (dark KITTI pixels, delta down to -32) gets a positive S and does not round-trip, exactly as
in the reference pipeline; the inverse scales H by 6 / 360 (a float multiply, not a division by 60).  Still a restatement: the
fixture is produced by the same formulas (oracle/gen_golden.py: _cv2_color_shim), so this ONE conversion is pinned to OpenCV's
published source, not to its binaries."""
import numpy as np

_PERMS = ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0))        # pd.py:143-145


_EPS = np.float32(1.1920929e-07)           # FLT_EPSILON


def bgr_to_hsv(img):
    """float32 [H, W, 3] with channels (B, G, R) in any range -> (H in [0, 360), S, V = max); OpenCV's RGB2HSV_f."""
    img = np.asarray(img, dtype=np.float32)
    b, g, r = img[..., 0], img[..., 1], img[..., 2]
    v = np.maximum(np.maximum(b, g), r)
    diff = (v - np.minimum(np.minimum(b, g), r)).astype(np.float32)
    s = (diff / (np.abs(v) + _EPS)).astype(np.float32)
    k = (60.0 / (diff + _EPS).astype(np.float64)).astype(np.float32)       # (float)(60. / (diff + FLT_EPSILON))
    h = np.where(v == r, (g - b) * k, np.where(v == g, (b - r) * k + np.float32(120), (r - g) * k + np.float32(240))).astype(np.float32)
    h = np.where(h < 0, h + np.float32(360), h).astype(np.float32)
    return np.stack([h, s, v], -1)


def hsv_to_bgr(img):
    """OpenCV's HSV2RGB_f for a 360-degree hue range: H * (6 / 360) wrapped into [0, 6), sector table."""
    img = np.asarray(img, dtype=np.float32)
    h, s, v = img[..., 0], img[..., 1], img[..., 2]
    hh = (h * np.float32(6.0 / 360.0)).astype(np.float32)
    hh = (hh - np.float32(6) * np.floor(hh / np.float32(6))).astype(np.float32)          # do h += 6 while (h < 0) / h -= 6 while (h >= 6)
    hh = np.where(hh >= 6, hh - np.float32(6), hh).astype(np.float32)                     # (the subtraction above can round up to 6)
    sector = np.floor(hh)
    f = (hh - sector).astype(np.float32)
    sector = sector.astype(np.int64) % 6
    p = v * (np.float32(1) - s)
    q = v * (np.float32(1) - s * f)
    t = v * (np.float32(1) - s * (np.float32(1) - f))
    # sector -> (r, g, b)   (OpenCV's sector_data table)
    r = np.choose(sector, [v, q, p, p, t, v])
    g = np.choose(sector, [t, v, v, q, p, p])
    b = np.choose(sector, [p, p, t, v, v, q])
    return np.stack([b, g, r], -1).astype(np.float32)


class PhotometricDistort:
    """``PhotometricDistort()(image float32 [H, W, 3]) -> image float32`` (pd.py:398-416)."""

    def __init__(self, brightness_delta=32.0, contrast=(0.5, 1.5), saturation=(0.5, 1.5), hue_delta=18.0):
        self.brightness_delta = brightness_delta
        self.contrast = contrast
        self.saturation = saturation
        self.hue_delta = hue_delta

    def _contrast(self, im):
        if np.random.randint(2):
            im *= np.random.uniform(*self.contrast)
        return im

    def __call__(self, image):
        rnd = np.random
        im = image.copy()
        if rnd.randint(2):                                               # RandomBrightness (pd.py:189-200)
            im += rnd.uniform(-self.brightness_delta, self.brightness_delta)
        contrast_first = bool(rnd.randint(2))
        if contrast_first:
            im = self._contrast(im)
        im = bgr_to_hsv(im)
        if rnd.randint(2):                                               # RandomSaturation (pd.py:114-126)
            im[:, :, 1] *= rnd.uniform(*self.saturation)
        if rnd.randint(2):                                               # RandomHue (pd.py:129-140)
            im[:, :, 0] += rnd.uniform(-self.hue_delta, self.hue_delta)
            im[:, :, 0][im[:, :, 0] > 360.0] -= 360.0
            im[:, :, 0][im[:, :, 0] < 0.0] += 360.0
        im = hsv_to_bgr(im)
        if not contrast_first:
            im = self._contrast(im)
        if rnd.randint(2):                                               # RandomLightingNoise (pd.py:143-154)
            im = im[:, :, _PERMS[rnd.randint(len(_PERMS))]]
        return im
