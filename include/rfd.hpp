// rfd.hpp -- C++17 host facade over the C ABI of librfd_hip.so (include/rfd.h), header-only.
//
// The reference's host code is a compiled Rust crate; no Rust toolchain exists in the build image, so this is the
// compiled-language mirror of the types a maintainer's Rust facade would expose (INTEGRATION.md shows that Rust side).
// Names, argument meaning and error behaviour follow the reference:
//   rfd::FaceDetectionConfig          <- FaceDetectionConfig::new            src/pipeline/face_pipeline/config.rs:13-33
//   rfd::RetinaFaceDetection          <- struct RetinaFaceDetection + ::new  src/pipeline/module/face_detection.rs:19-129
//   RetinaFaceDetection::call         <- ::call(&self, image:&Mat, is_debug) face_detection.rs:496-513
//                                        -> (Array2<f32> [K,5], Option<Array3<f32>> [K,5,2])
//   rfd::FaceSelection::call          <- FaceSelection::call                 src/pipeline/module/face_selection.rs:72-189
//   rfd::FaceAlignment::call          <- FaceAlignment::call                 src/pipeline/module/face_alignment.rs:27-141
//   rfd::Error                        <- anyhow::Error: every failing entry point throws it (status + message)
// What changed against the reference constructor: the Triton client / model config / model name arguments are gone
// (the network runs in-process); device_id, max_det and the backbone are new.
#ifndef RFD_HPP
#define RFD_HPP

#include <array>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rfd.h"

namespace rfd {

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string &what) : std::runtime_error("rfd status " + std::to_string(st) + ": " + what), status(st) {}
};
inline void check(int status)
{
    if (status < 0) throw Error(status, rfd_last_error() ? rfd_last_error() : "");
}

// config.rs:13-33
struct FaceDetectionConfig {
    std::string model_name = "face_detection_retina"; // kept for source compatibility; unused (no Triton)
    std::pair<int, int> image_size{640, 640};          // (w, h)
    int max_batch_size = 1;
    float confidence_threshold = 0.7f;
    float iou_threshold = 0.45f;
    int timeout = 20;                                  // unused (no RPC)
};

// the decoded frame the reference passes around as opencv::core::Mat (CV_8UC3, BGR; utils.rs:8-52)
struct Mat {
    const uint8_t *data = nullptr;
    int rows = 0, cols = 0;
    std::ptrdiff_t step = 0; // bytes per row
    int channels = 3;
    Mat() = default;
    Mat(const uint8_t *d, int r, int c, std::ptrdiff_t s = 0, int ch = 3) : data(d), rows(r), cols(c), step(s ? s : (std::ptrdiff_t)c * ch), channels(ch) {}
};

// (Array2<f32> [K,5], Array3<f32> [K,5,2]) of face_detection.rs:496, row-major
struct Detections {
    std::size_t k = 0;
    std::vector<float> det; // k x 5: x1, y1, x2, y2, score in source-image pixels, descending score
    std::vector<float> kps; // k x 5 x 2
    const float *box(std::size_t i) const { return det.data() + 5 * i; }
    const float *landmarks(std::size_t i) const { return kps.data() + 10 * i; }
};

class RetinaFaceDetection {
  public:
    // RetinaFaceDetection::new(client, model_cfg, model_name, image_size, max_batch_size, conf, iou) minus the Triton
    // arguments (face_detection.rs:41-49)
    RetinaFaceDetection(std::pair<int, int> image_size, int max_batch_size, float confidence_threshold, float iou_threshold,
                        int device_id = 0, int max_det = 1024, int backbone = RFD_BACKBONE_R50, int precision = RFD_PRECISION_BF16)
    {
        rfd_config cfg;
        rfd_config_default(&cfg);
        cfg.image_w = image_size.first; cfg.image_h = image_size.second;
        cfg.max_batch_size = max_batch_size;
        cfg.confidence_threshold = confidence_threshold;
        cfg.iou_threshold = iou_threshold;
        cfg.device_id = device_id;
        cfg.max_det = max_det;
        cfg.backbone = backbone;
        cfg.precision = precision; // RFD_PRECISION_F32: the f32 parity mode (the reference's FP32 tensor contract, face_detection.rs:261)
        check(rfd_create(&cfg, &ctx_));
        max_det_ = (std::size_t)max_det;
        max_batch_ = max_batch_size;
    }
    explicit RetinaFaceDetection(const FaceDetectionConfig &c, int device_id = 0, int max_det = 1024, int backbone = RFD_BACKBONE_R50)
        : RetinaFaceDetection(c.image_size, c.max_batch_size, c.confidence_threshold, c.iou_threshold, device_id, max_det, backbone) {}
    RetinaFaceDetection(const RetinaFaceDetection &) = delete;
    RetinaFaceDetection &operator=(const RetinaFaceDetection &) = delete;
    ~RetinaFaceDetection() { rfd_destroy(ctx_); }

    void init_synthetic_weights(uint64_t seed) { check(rfd_init_synthetic_weights(ctx_, seed)); }
    void load_weights(const std::string &path) { check(rfd_load_weights(ctx_, path.c_str())); }
    void save_weights(const std::string &path) { check(rfd_save_weights(ctx_, path.c_str())); }

    // RetinaFaceDetection::call (face_detection.rs:496).  A non-3-channel image is an error, as in the reference
    // (at_2d::<Vec3b> fails, face_detection.rs:226).  Empty result: k = 0 (reference: (0,5) and (0,5,2), :413-419).
    Detections call(const Mat &image, std::optional<bool> /*is_debug*/ = std::nullopt) { return call_batch({image}).at(0); }

    // batch form: one call for n frames (the throughput entry; n <= max_batch_size)
    std::vector<Detections> call_batch(const std::vector<Mat> &images)
    {
        const int n = (int)images.size();
        std::vector<rfd_image> im(images.size());
        for (int i = 0; i < n; ++i) {
            if (images[i].channels != 3) throw Error(RFD_ERR_INVALID_ARG, "face_detection - expected a CV_8UC3 image");
            im[i] = rfd_image{images[i].data, images[i].rows, images[i].cols, images[i].step};
        }
        std::vector<float> boxes((std::size_t)n * max_det_ * 5), lmk((std::size_t)n * max_det_ * 10);
        std::vector<int32_t> count(n), total(n);
        rfd_dets out{boxes.data(), lmk.data(), count.data(), total.data()};
        check(rfd_detect_batch(ctx_, im.data(), n, &out));
        std::vector<Detections> res(n);
        for (int i = 0; i < n; ++i) {
            const std::size_t k = (std::size_t)count[i];
            res[i].k = k;
            res[i].det.assign(boxes.begin() + (std::size_t)i * max_det_ * 5, boxes.begin() + (std::size_t)i * max_det_ * 5 + k * 5);
            res[i].kps.assign(lmk.begin() + (std::size_t)i * max_det_ * 10, lmk.begin() + (std::size_t)i * max_det_ * 10 + k * 10);
        }
        return res;
    }

    rfd_ctx *raw() const { return ctx_; }
    std::size_t max_det() const { return max_det_; }

  private:
    rfd_ctx *ctx_ = nullptr;
    std::size_t max_det_ = 0;
    int max_batch_ = 0;
};

// FaceSelectionConfig::new (config.rs:98-117) + FaceSelection::call (face_selection.rs:72-189) on the device
struct FaceSelectionConfig {
    float margin_center_left_ratio = 0.3f, margin_center_right_ratio = 0.3f, margin_edge_ratio = 0.1f, minimum_face_ratio = 0.0075f;
};
struct SelectedFace {
    std::optional<std::array<float, 5>> bbox;                // None: no face passed the filters
    std::optional<std::array<float, 10>> key_points;         // None: no detection within 2 px of the chosen box
};
class FaceSelection {
  public:
    explicit FaceSelection(FaceSelectionConfig c = {}) : cfg_(c) {}
    // call(&self, img, bboxes, kps, is_enroll) -> (Option<bbox>, Option<kps>): here over the detector's output
    SelectedFace call(RetinaFaceDetection &det, const Mat &image, const Detections &d, bool is_enroll = false) const
    {
        const std::size_t md = det.max_det();
        std::vector<float> boxes(md * 5, 0.f), lmk(md * 10, 0.f);
        std::copy(d.det.begin(), d.det.end(), boxes.begin());
        std::copy(d.kps.begin(), d.kps.end(), lmk.begin());
        int32_t count = (int32_t)d.k, total = (int32_t)d.k;
        rfd_dets in{boxes.data(), lmk.data(), &count, &total};
        rfd_selection_config sc{cfg_.margin_center_left_ratio, cfg_.margin_center_right_ratio, cfg_.margin_edge_ratio, cfg_.minimum_face_ratio};
        const int h = image.rows, w = image.cols;
        float ob[5], ok[10];
        int32_t found = 0;
        check(rfd_select_faces(det.raw(), &in, &h, &w, 1, &sc, is_enroll ? 1 : 0, ob, ok, &found));
        SelectedFace s;
        if (found & 1) { s.bbox.emplace(); std::copy(ob, ob + 5, s.bbox->begin()); }
        if (found == 3) { s.key_points.emplace(); std::copy(ok, ok + 10, s.key_points->begin()); }
        return s;
    }

  private:
    FaceSelectionConfig cfg_;
};

// FaceAlignmentConfig::new (config.rs:37-56) + FaceAlignment::call (face_alignment.rs:27-141) on the device
class FaceAlignment {
  public:
    FaceAlignment() { rfd_alignment_config_default(&cfg_); }
    FaceAlignment(std::pair<int, int> image_size, const std::array<float, 10> &standard_landmarks)
    {
        rfd_alignment_config_default(&cfg_);
        cfg_.out_w = image_size.first; cfg_.out_h = image_size.second;
        std::copy(standard_landmarks.begin(), standard_landmarks.end(), cfg_.standard_landmarks);
    }
    // -> the aligned out_h x out_w x 3 u8 BGR crop (the reference returns a Mat); throws where the reference returns Err
    std::vector<uint8_t> call(RetinaFaceDetection &det, const Mat &img, const SelectedFace &face) const
    {
        rfd_image im{img.data, img.rows, img.cols, img.step};
        float box[5] = {0, 0, 0, 0, 0}, kps[10] = {0};
        int32_t found = 0, status = 0;
        if (face.bbox) { std::copy(face.bbox->begin(), face.bbox->end(), box); found |= 1; }
        if (face.key_points) { std::copy(face.key_points->begin(), face.key_points->end(), kps); found |= 2; }
        std::vector<uint8_t> crop((std::size_t)cfg_.out_w * cfg_.out_h * 3);
        check(rfd_align_faces(det.raw(), &im, 1, box, kps, &found, &cfg_, crop.data(), &status));
        if (status < 0) throw Error(RFD_ERR_INVALID_ARG, "face_alignment - no key points / no face / crop outside the image (status " + std::to_string(status) + ")");
        return crop;
    }

  private:
    rfd_alignment_config cfg_;
};

} // namespace rfd
#endif
