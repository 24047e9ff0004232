"""The convergence rule of generate_pixel (raytracer_core/src/lib.rs:276-313) as a host-side state machine over a sequence
of per-pass 8-bit images: what the status / same_color_count of every pixel must be after each pass. Test helper."""
import numpy as np

NOT_STARTED, UNSTABLE, FINAL = 0, 2, 3


class RuleEmulator:
    def __init__(self, h, w):
        self.status = np.zeros((h, w), np.uint8)
        self.count = np.zeros((h, w), np.uint8)
        self.last = np.zeros((h, w, 3), np.uint8)

    def live(self):
        return self.status != FINAL

    def step(self, rgb):
        """rgb: (h, w, 3) colours of this pass (only read where the pixel was not Final before the pass)"""
        live = self.status != FINAL                 # lib.rs:277-279
        has_last = self.status != NOT_STARTED
        same = np.all(rgb == self.last, axis=2)
        cnt = np.where(has_last, np.where(same, self.count + 1, 0), self.count).astype(np.uint8)   # lib.rs:300-305
        self.count = np.where(live, cnt, self.count).astype(np.uint8)
        st = np.where(self.count > 3, FINAL, UNSTABLE).astype(np.uint8)                               # lib.rs:306-311
        self.status = np.where(live, st, self.status).astype(np.uint8)
        self.last = np.where(live[..., None], rgb, self.last).astype(np.uint8)
