"""Python facade with the reference's surface (fast_slic/base_slic.py:3-59) on top of the HIP engine.

The `BaseSlic` class below keeps the reference's constructor, attributes and `iterate()` body as they are (north_star: "keeping
the fast_slic.Slic.iterate() Python surface"; SURVEY.md section 2 #1 "keep verbatim"), so that code written against
`fast_slic.Slic` runs unchanged.  That part derives from fast_slic/base_slic.py of Algy/fast-slic:

    MIT License -- Copyright (c) 2019 Alchan Kim

    Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated documentation
    files (the "Software"), to deal in the Software without restriction, including without limitation the rights to use, copy,
    modify, merge, publish, distribute, sublicense, and/or sell copies of the Software, and to permit persons to whom the Software
    is furnished to do so, subject to the following conditions: The above copyright notice and this permission notice shall be
    included in all copies or substantial portions of the Software.  THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY
    KIND, EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED TO THE WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND
    NONINFRINGEMENT.  IN NO EVENT SHALL THE AUTHORS OR COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY,
    WHETHER IN AN ACTION OF CONTRACT, TORT OR OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR
    OTHER DEALINGS IN THE SOFTWARE.
"""
from ._binding import SlicModel, ARCH_NAME


class BaseSlic(object):
    arch_name = "__TODO__"

    def __init__(self,
                 num_components=400,
                 slic_model=None,
                 compactness=10,
                 min_size_factor=0.25,
                 subsample_stride=3,
                 convert_to_lab=True,
                 preemptive=False,
                 preemptive_thres=0.05,
                 manhattan_spatial_dist=True,
                 debug_mode=False,
                 num_threads=-1):
        self.compactness = compactness
        self.subsample_stride = subsample_stride
        self.min_size_factor = min_size_factor
        self._slic_model = slic_model and slic_model.copy() or self.make_slic_model(num_components)
        self._last_assignment = None

        self.convert_to_lab = convert_to_lab
        self._slic_model.preemptive = preemptive
        self._slic_model.preemptive_thres = preemptive_thres
        self._slic_model.manhattan_spatial_dist = manhattan_spatial_dist
        self._slic_model.num_threads = num_threads
        self._slic_model.debug_mode = debug_mode

    @property
    def convert_to_lab(self):
        return self._slic_model.convert_to_lab

    @convert_to_lab.setter
    def convert_to_lab(self, v):
        self._slic_model.convert_to_lab = v

    @property
    def slic_model(self):
        return self._slic_model

    @property
    def last_assignment(self):
        return self._last_assignment

    def iterate(self, image, max_iter=10):
        if not self._slic_model.initialized:
            self._slic_model.initialize(image)
        assignment = self._slic_model.iterate(image, max_iter, self.compactness, self.min_size_factor,
                                              self.subsample_stride)
        self._last_assignment = assignment
        return assignment

    @property
    def num_components(self):
        return self._slic_model.num_components

    def make_slic_model(self, num_components):
        return SlicModel(num_components, self.arch_name)


class SlicHip(BaseSlic):
    """Drop-in for fast_slic.Slic / fast_slic.avx2.SlicAvx2 (fast_slic/avx2.py:10-11) on MI355X."""
    arch_name = ARCH_NAME


class SlicRealDistHip(BaseSlic):
    """fast_slic.SlicRealDist (fast_slic/base_slic.py:63-71): f32 distances, quantised centroids."""
    arch_name = ARCH_NAME
    real_dist_type = "standard"

    def make_slic_model(self, num_components):
        model = SlicModel(num_components, self.arch_name)
        model.real_dist = True
        model.real_dist_type = self.real_dist_type
        return model


class SlicRealDistL2Hip(SlicRealDistHip):
    """fast_slic.SlicRealDistL2 (fast_slic/base_slic.py:73-75): squared colour and spatial terms."""
    real_dist_type = "l2"


class SlicRealDistNoQHip(SlicRealDistHip):
    """fast_slic.SlicRealDistNoQ (fast_slic/base_slic.py:77-85): float (non-quantised) centroids."""
    real_dist_type = "noq"

    def __init__(self, *args, **kwargs):
        float_color = kwargs.pop("float_color", True)
        super(SlicRealDistNoQHip, self).__init__(*args, **kwargs)
        self._slic_model.float_color = float_color      # stored and unused, as in the reference (src/context.h:116)


class LSCHip(SlicRealDistHip):
    """Drop-in for fast_slic.LSC / fast_slic.avx2.LSCAvx2 (fast_slic/base_slic.py:87-89; avx2.py:13-14)."""
    real_dist_type = "lsc"
