// libstillleben_diff_python -- the reference's second extension module (python/src/bridge_diff.cpp:160-180: exactly these two
// functions, imported by python/stillleben/diff.py:22-30) as host C++ over the C-ABI of libslhip.so: argument checks and error
// texts follow bridge_diff.cpp:13-31, 71-96; the stencils themselves run in slhip_diff_sobel_valid / slhip_diff_dilate
// (csrc/slhip_diff.hip, CPU-loop semantics of bridge_diff.cpp:37-69, 102-157).  The reference dispatches on the device of
// the first argument; here every tensor goes through the HIP device (there is no CPU path in this product) and the results
// come back on the device of the first argument.
//
// Built by __graft_entry__.build() with g++ against torch's headers (pybind11 + ATen) and linked to libslhip.so;
// imported as stillleben.lib.libstillleben_diff_python (and stillleben_amd.lib...).
#include <torch/extension.h>

#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>

#include <stdexcept>
#include <tuple>

#include "slhip.h"

namespace {

void check(int status, const char* what)
{
    if (status != 0) {
        const char* msg = slhip_last_error();
        throw std::runtime_error(std::string(what) + " failed: " + (msg ? msg : "?"));
    }
}

c10::Device hip_device(const at::Tensor& first)
{
    if (first.is_cuda()) return first.device();
    if (!torch::cuda::is_available()) throw std::runtime_error("libstillleben_diff_python: no HIP device (there is no CPU path)");
    return c10::Device(c10::kCUDA, c10::hip::current_device());
}

at::Tensor generateSobelValidMask(at::Tensor instance_indices, at::Tensor depth_image)
{
    if (instance_indices.dim() != 2 || depth_image.dim() != 2)
        throw std::invalid_argument{"input tensors should be two-dimensional"};
    if (instance_indices.size(0) != depth_image.size(0) || instance_indices.size(1) != depth_image.size(1))
        throw std::invalid_argument{"instance_indices and depth_image should be of same height and width"};
    const auto out_dev = instance_indices.device();
    const auto dev = hip_device(instance_indices);
    const int64_t H = instance_indices.size(0), W = instance_indices.size(1);
    at::Tensor inst = instance_indices.to(dev, at::kShort).contiguous();
    at::Tensor depth = depth_image.to(dev, at::kFloat).contiguous();
    at::Tensor valid = at::empty({H, W}, at::TensorOptions().dtype(at::kByte).device(dev));
    c10::hip::HIPGuard guard(dev.index());
    check(slhip_diff_sobel_valid(inst.data_ptr<int16_t>(), depth.data_ptr<float>(), 1, (uint32_t)H, (uint32_t)W,
                                 valid.data_ptr<uint8_t>(), c10::hip::getCurrentHIPStream(dev.index()).stream()),
          "slhip_diff_sobel_valid");
    return valid.to(at::kBool).to(out_dev);
}

std::tuple<at::Tensor, at::Tensor> dilateObjectMask(at::Tensor object_mask, at::Tensor sobel_valid_mask, at::Tensor coordinates)
{
    if (object_mask.dim() != 2 || sobel_valid_mask.dim() != 2)
        throw std::invalid_argument{"object_mask &  sobel_valid_mask should be two-dimensional"};
    if (coordinates.dim() != 3)
        throw std::invalid_argument{"coordinates should be three-dimensional"};
    if (object_mask.size(0) != sobel_valid_mask.size(0) || object_mask.size(1) != sobel_valid_mask.size(1) ||
        object_mask.size(0) != coordinates.size(0) || object_mask.size(1) != coordinates.size(1))
        throw std::invalid_argument{"object_mask, sobel_valid_mask, and coordinates should be of same height and width"};
    const auto out_dev = object_mask.device();
    const auto dev = hip_device(object_mask);
    const int64_t H = object_mask.size(0), W = object_mask.size(1), C = coordinates.size(2);
    at::Tensor m = object_mask.to(dev, at::kByte).contiguous();
    at::Tensor v = sobel_valid_mask.to(dev, at::kByte).contiguous();
    at::Tensor c = coordinates.to(dev, at::kFloat).contiguous();
    at::Tensor om = at::empty({H, W}, at::TensorOptions().dtype(at::kByte).device(dev));
    at::Tensor oc = at::empty({H, W, 3}, at::TensorOptions().dtype(at::kFloat).device(dev));
    c10::hip::HIPGuard guard(dev.index());
    check(slhip_diff_dilate(m.data_ptr<uint8_t>(), v.data_ptr<uint8_t>(), c.data_ptr<float>(), (uint32_t)C, (uint32_t)H, (uint32_t)W,
                            om.data_ptr<uint8_t>(), oc.data_ptr<float>(), c10::hip::getCurrentHIPStream(dev.index()).stream()),
          "slhip_diff_dilate");
    return std::make_tuple(om.to(at::kBool).to(out_dev), oc.to(out_dev));
}

}  // namespace

PYBIND11_MODULE(libstillleben_diff_python, m)
{
    m.def("generate_sobel_valid_mask", generateSobelValidMask,
          R"EOS(
            Generate mask of valid pixels.

            :param instance_indices: HxW short tensor with instance indices
            :param depth_image: HxW float tensor with depth
            :return: HxW bool tensor

            The returned mask is unset iff the pixel is close to an occluder
            (i.e. there is a neighboring pixel of another object that is closer).
        )EOS");
    m.def("dilate_object_mask", dilateObjectMask,
          R"EOS(
            Dilate object mask.
        )EOS");
}
