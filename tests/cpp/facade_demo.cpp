// facade_demo.cpp -- compiled-language user of librfd_hip.so through include/rfd.hpp (what the Rust crate's
// FacePipeline::extract does, pipeline.rs:198-216): detect -> select -> align on a raw BGR frame read from a file.
// Prints every number with %.9g / as bytes so that tests/test_cpp_facade_gpu.py can compare bit-for-bit with the
// Python binding.   usage: facade_demo <frame.raw> <h> <w> <backbone 0|1> <seed> [crop.out]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>

#include "rfd.hpp"

int main(int argc, char **argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage: %s frame.raw h w backbone seed [crop.out]\n", argv[0]); return 2; }
    const int h = std::atoi(argv[2]), w = std::atoi(argv[3]), backbone = std::atoi(argv[4]);
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> px((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if ((long long)px.size() != (long long)h * w * 3) { std::fprintf(stderr, "frame size mismatch\n"); return 2; }
    try {
        rfd::FaceDetectionConfig cfg;                 // 640x640, 0.7, 0.45 (config.rs:23-32)
        cfg.confidence_threshold = 0.3f;              // synthetic weights: keep some detections
        rfd::RetinaFaceDetection det(cfg, 0, 512, backbone);
        det.init_synthetic_weights((uint64_t)std::atoll(argv[5]));
        const rfd::Mat image(px.data(), h, w);
        const rfd::Detections d = det.call(image);
        std::printf("K %zu\n", d.k);
        for (std::size_t i = 0; i < d.k; ++i) {
            std::printf("det");
            for (int j = 0; j < 5; ++j) std::printf(" %.9g", d.box(i)[j]);
            std::printf(" kps");
            for (int j = 0; j < 10; ++j) std::printf(" %.9g", d.landmarks(i)[j]);
            std::printf("\n");
        }
        const rfd::SelectedFace face = rfd::FaceSelection().call(det, image, d);
        std::printf("selected %d %d\n", face.bbox ? 1 : 0, face.key_points ? 1 : 0);
        if (face.bbox) {
            std::printf("bbox");
            for (float v : *face.bbox) std::printf(" %.9g", v);
            std::printf("\n");
        }
        if (face.bbox && face.key_points && argc > 6) {
            const std::vector<uint8_t> crop = rfd::FaceAlignment().call(det, image, face);
            std::ofstream o(argv[6], std::ios::binary);
            o.write((const char *)crop.data(), (std::streamsize)crop.size());
            std::printf("crop %zu\n", crop.size());
        }
        // error behaviour: a 1-channel Mat is rejected like the reference's at_2d::<Vec3b> failure
        try {
            rfd::Mat gray(px.data(), h, w, w, 1);
            det.call(gray);
            std::printf("gray accepted\n");
        } catch (const rfd::Error &e) {
            std::printf("gray rejected %d\n", e.status);
        }
    } catch (const rfd::Error &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
