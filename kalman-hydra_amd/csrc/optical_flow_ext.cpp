// optical_flow_ext <video.npy> <output prefix> [alpha] [gamma] [scale_factor] [inner_it] [outer_it] [solver_it]
//
// The reference's flow tool (reference src/optical_flow_ext.cpp:333-416 process(), :441-507 main()) as a native
// program over the C-ABI of libhydra_mi.so: for every consecutive frame pair of the video the Brox flow, written as
// <prefix>_%03d_x.mat / <prefix>_%03d_y.mat in the reference's .mat format (:47-108: int32 type = 5 (CV_32FC1),
// int32 width, int32 height, then width*height little-endian f32, row-major).  Same positional arguments and defaults
// (:453-488) and the same messages on bad input (exit code 1).  What differs: the frame source.  There is no OpenCV on
// this path, so the video is a NumPy .npy file -- uint8, C order, shape (frames, H, W) or
// (frames, H, W, 3) in BGR -- read frame by frame; BGR is converted as cvtColor(BGR2GRAY) does (:366-368):
// rint(0.114 B + 0.587 G + 0.299 R).  The pairs are independent (the loop of process() carries nothing but the
// previous frame), so they go to the GPU in series of HYDRA_MI_FLOW_BATCH (default 16) pairs: hm_brox_calc_batch.
// kalman-hydra_amd/../optical_flow_ext.py is the same tool in Python; the two write identical files (tests).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/hydra_mi.h"

namespace {

struct NpyVideo {
    FILE *f = nullptr;
    long data_start = 0;
    int frames = 0, H = 0, W = 0, chans = 1;
    ~NpyVideo() { if (f) fclose(f); }

    // the header of a .npy file, format versions 1.0 - 3.0: magic, version, little-endian header length, a Python dict
    bool open(const std::string &path, std::string &why)
    {
        f = fopen(path.c_str(), "rb");
        if (!f) { why = "cannot open " + path; return false; }
        unsigned char head[12];
        if (fread(head, 1, 10, f) != 10 || memcmp(head, "\x93NUMPY", 6) != 0) { why = path + " is not a .npy file"; return false; }
        size_t hlen = head[8] | (head[9] << 8);
        if (head[6] >= 2) {
            if (fread(head + 10, 1, 2, f) != 2) { why = "truncated header"; return false; }
            hlen |= ((size_t)head[10] << 16) | ((size_t)head[11] << 24);
        }
        std::string dict(hlen, '\0');
        if (fread(&dict[0], 1, hlen, f) != hlen) { why = "truncated header"; return false; }
        data_start = ftell(f);
        if (dict.find("'|u1'") == std::string::npos && dict.find("'<u1'") == std::string::npos && dict.find("'uint8'") == std::string::npos) {
            why = "expected an 8-bit (uint8) array";
            return false;
        }
        if (dict.find("'fortran_order': False") == std::string::npos) { why = "expected a C-order array"; return false; }
        const size_t s0 = dict.find("'shape':");
        const size_t p0 = s0 == std::string::npos ? s0 : dict.find('(', s0), p1 = p0 == std::string::npos ? p0 : dict.find(')', p0);
        if (p1 == std::string::npos) { why = "no shape in the header"; return false; }
        std::vector<long> dims;
        std::istringstream ss(dict.substr(p0 + 1, p1 - p0 - 1));
        std::string tok;
        while (std::getline(ss, tok, ',')) {
            bool digits = false;
            for (char ch : tok) digits = digits || (ch >= '0' && ch <= '9');
            if (digits) dims.push_back(atol(tok.c_str()));
        }
        if (!(dims.size() == 3 || (dims.size() == 4 && dims[3] == 3))) { why = "expected an array of shape (frames, H, W[, 3])"; return false; }
        frames = (int)dims[0]; H = (int)dims[1]; W = (int)dims[2]; chans = dims.size() == 4 ? 3 : 1;
        if (frames < 1 || H < 1 || W < 1) { why = "empty video"; return false; }
        return true;
    }

    // frame k as 8-bit gray (H*W bytes)
    bool gray(int k, uint8_t *out, std::vector<uint8_t> &tmp)
    {
        const size_t px = (size_t)H * W;
        if (fseek(f, data_start + (long)((size_t)k * px * chans), SEEK_SET) != 0) return false;
        if (chans == 1) return fread(out, 1, px, f) == px;
        tmp.resize(px * 3);
        if (fread(tmp.data(), 1, px * 3, f) != px * 3) return false;
        for (size_t i = 0; i < px; i++) {
            const double g = 0.114 * (double)tmp[3 * i] + 0.587 * (double)tmp[3 * i + 1] + 0.299 * (double)tmp[3 * i + 2];
            out[i] = (uint8_t)nearbyint(g);                   // np.rint: ties to even
        }
        return true;
    }
};

// writeMatToFile of the reference for a CV_32FC1 matrix (src/optical_flow_ext.cpp:47-79)
bool write_mat_f32(const std::string &path, const float *a, int width, int height)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const int32_t head[3] = {5, width, height};
    bool ok = fwrite(head, sizeof(int32_t), 3, f) == 3;
    ok = ok && fwrite(a, sizeof(float), (size_t)width * height, f) == (size_t)width * height;
    return fclose(f) == 0 && ok;
}

void help(const char *me)
{
    std::cout << "This program reads the frames of a video (.npy: uint8, shape (frames, H, W) or (frames, H, W, 3) BGR)." << std::endl
              << "It then uses the Brox implementation of libhydra_mi.so (MI355X) to compute optic flow" << std::endl << std::endl
              << "Usage:\n" << me << " <video .npy> <output filename> [alpha] [gamma] [scale_factor] [inner_it] [outer_it] [solver_it] " << std::endl
              << "Parameters:" << std::endl
              << "  float alpha = smoothness regularization parameter -- higher = more smooth" << std::endl
              << "  float gamma = gradient constancy importance -- higher = greater gradient importance" << std::endl
              << "  float scale_factor = pyramid scale factor = ratio between pyramid scales" << std::endl
              << "  int inner_iterations = number of lagged non-linearity iterations (inner loop)" << std::endl
              << "  int outer_iterations = number of warping iterations (number of pyramid levels)" << std::endl
              << "  int solver_iterations = number of linear system solver iterations" << std::endl
              << "Output: <output filename>_%03d_x.mat and _y.mat for every consecutive frame pair" << std::endl;
}

template <typename T>
T arg_or(int ac, char **av, int i, T dflt)
{
    if (ac <= i) return dflt;
    T v = dflt;
    std::istringstream iss(av[i]);
    iss >> v;
    return v;
}

}  // namespace

int main(int ac, char **av)
{
    if (ac < 3) {
        help(av[0]);
        return 1;
    }
    const std::string arg = av[1], fn_out = av[2];
    const float alpha = arg_or<float>(ac, av, 3, 0.197f), gamma = arg_or<float>(ac, av, 4, 50.0f), scale_factor = arg_or<float>(ac, av, 5, 0.8f);
    const int inner_it = arg_or<int>(ac, av, 6, 10), outer_it = arg_or<int>(ac, av, 7, 77), solver_it = arg_or<int>(ac, av, 8, 10);
    std::cout << "Using Brox optic flow parameters:" << std::endl
              << "   alpha = " << alpha << " = smoothness regularization parameter" << std::endl
              << "   gamma = " << gamma << " = gradient constancy importance" << std::endl
              << "   scale_factor = " << scale_factor << " = pyramid scale factor" << std::endl
              << "   inner_iterations = " << inner_it << " = number of lagged non-linearity iterations (inner loop)" << std::endl
              << "   outer_iterations = " << outer_it << " = number of warping iterations (number of pyramid levels)" << std::endl
              << "   solver_iterations = " << solver_it << " = number of linear system solver iterations" << std::endl;
    NpyVideo video;
    std::string why;
    if (!video.open(arg, why)) {
        std::cerr << "Failed to open the video file: " << why << "\n" << std::endl;
        help(av[0]);
        return 1;
    }
    const int pairs = video.frames - 1;
    int B = 16;
    if (const char *e = getenv("HYDRA_MI_FLOW_BATCH")) B = atoi(e) > 0 ? atoi(e) : B;
    if (B > pairs) B = pairs > 0 ? pairs : 1;
    hm_brox_t h = nullptr;
    if (hm_brox_create(0, video.W, video.H, B, alpha, gamma, scale_factor, inner_it, outer_it, solver_it, &h) != HM_OK) {
        std::cerr << "hm_brox_create failed: " << hm_last_error() << std::endl;
        return 1;
    }
    const size_t px = (size_t)video.W * video.H;
    std::vector<uint8_t> frames((size_t)(B + 1) * px), tmp;
    std::vector<float> fx((size_t)B * px), fy((size_t)B * px);
    int rc = 0;
    for (int s = 0; s < pairs && rc == 0; s += B) {
        const int nb = pairs - s < B ? pairs - s : B;
        // frames s .. s + nb, contiguous: frame0 of pair i is frame s + i, frame1 is the one after it
        for (int k = 0; k <= nb && rc == 0; k++)
            if (!video.gray(s + k, frames.data() + (size_t)k * px, tmp)) { std::cerr << "Failed to read frame " << s + k << std::endl; rc = 1; }
        if (rc) break;
        if (hm_brox_calc_batch(h, nb, frames.data(), frames.data() + px, fx.data(), fy.data()) != HM_OK) {
            std::cerr << "hm_brox_calc_batch failed: " << hm_last_error() << std::endl;
            rc = 1;
            break;
        }
        for (int k = 0; k < nb && rc == 0; k++) {
            char count[16];
            snprintf(count, sizeof count, "%03d", s + k);
            const std::string base = fn_out + "_" + count;
            if (!write_mat_f32(base + "_x.mat", fx.data() + (size_t)k * px, video.W, video.H) ||
                !write_mat_f32(base + "_y.mat", fy.data() + (size_t)k * px, video.W, video.H)) {
                std::cerr << "File I/O error" << std::endl;      // the reference's message (:60, :120)
                rc = 1;
            }
        }
    }
    hm_brox_destroy(h);
    if (rc == 0) std::cout << "Finished." << std::endl;
    return rc;
}
