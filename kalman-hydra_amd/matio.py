"""The flow tool's ``.mat`` wire format (not MATLAB's).

reference src/optical_flow_ext.cpp:47-108 (writeMatToFile) and :110-170
(readFileToMat): little-endian ``int32 type; int32 width; int32 height`` followed
by ``width*height`` elements in row-major order.  ``type`` is the OpenCV type code:
5 = CV_32FC1 (f32), 6 = CV_64FC1 (f64), 21 = CV_32FC3, 22 = CV_64FC3.  Flow planes
are written as type 5 under the names ``<prefix>_%03d_x.mat`` / ``_y.mat``
(:362-364, :408-409), index 0 being the pair (frame 0, frame 1).
"""
import struct

import numpy as np

CV_32FC1, CV_64FC1, CV_32FC3, CV_64FC3 = 5, 6, 21, 22
_TYPES = {CV_32FC1: (np.float32, 1), CV_64FC1: (np.float64, 1), CV_32FC3: (np.float32, 3), CV_64FC3: (np.float64, 3)}


def write_mat(path, a):
    a = np.asarray(a)
    if a.ndim == 2:
        chans = 1
    elif a.ndim == 3 and a.shape[2] == 3:
        chans = 3
    else:
        raise ValueError("write_mat: array must be HxW or HxWx3")
    if a.dtype == np.float32:
        code = CV_32FC1 if chans == 1 else CV_32FC3
    elif a.dtype == np.float64:
        code = CV_64FC1 if chans == 1 else CV_64FC3
    else:
        raise ValueError("write_mat: wrong Mat type: must be CV_32F, CV_64F, CV_32FC3 or CV_64FC3")
    with open(path, "wb") as f:
        f.write(struct.pack("<iii", code, a.shape[1], a.shape[0]))
        f.write(np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<"), copy=False).tobytes())


def read_mat(path):
    """-> array of shape (height, width) or (height, width, 3); IOError if the file is missing/short."""
    with open(path, "rb") as f:
        head = f.read(12)
        if len(head) != 12:
            raise IOError("%s: truncated header" % path)
        code, width, height = struct.unpack("<iii", head)
        if code not in _TYPES:
            raise IOError("%s: wrong Mat type %d: must be CV_32F, CV_64F, CV_32FC3 or CV_64FC3" % (path, code))
        dtype, chans = _TYPES[code]
        n = width * height * chans
        data = np.frombuffer(f.read(n * np.dtype(dtype).itemsize), dtype=np.dtype(dtype).newbyteorder("<"))
        if data.size != n:
            raise IOError("%s: truncated data" % path)
    data = data.astype(dtype)
    return data.reshape(height, width) if chans == 1 else data.reshape(height, width, 3)


def flow_names(prefix, index):
    return "%s_%03d_x.mat" % (prefix, index), "%s_%03d_y.mat" % (prefix, index)


def write_flow(prefix, index, flowx, flowy):
    fx, fy = flow_names(prefix, index)
    write_mat(fx, np.asarray(flowx, np.float32))
    write_mat(fy, np.asarray(flowy, np.float32))
