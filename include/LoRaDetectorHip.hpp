// LoRaDetectorHip.hpp -- header-only C++ wrapper with the member signatures of the reference's
// `template <typename Type> class LoRaDetector` (LoRaDetector.hpp:8-72), backed by the C ABI of
// liblorahip.so (include/lorahip.h, level 1). LoRaDemod.cpp swaps
//     LoRaDetector<float> _detector;          (LoRaDemod.cpp:364)
// for
//     LoRaDetectorHip<float> _detector;
// and nothing else in the block changes: feed() stores one sample, detect()
// returns the arg-max bin and writes power / powerAvg / fIndex and, if asked, the N FFT bins.
//
// Differences from the reference class, all at construction time: the constructor throws
// std::runtime_error when there is no gfx950 device or N is not a supported power of two
// (the reference accepts any N; LoRaDemod only ever passes 1 << sf). There is no CPU fallback.
#pragma once
#include <complex>
#include <cstddef>
#include <stdexcept>
#include <string>
#include <type_traits>
#include "lorahip.h"

template <typename Type>
class LoRaDetectorHip
{
    static_assert(std::is_same<Type, float>::value, "the HIP path computes in fp32, like LoRaDemod's LoRaDetector<float>");

public:
    explicit LoRaDetectorHip(const size_t N, const int device = 0) : N(N), _det(nullptr)   // LoRaDetector.hpp:12
    {
        const int rc = lorahip_detector_create(&_det, device, N);
        if (rc != LORAHIP_OK)
            throw std::runtime_error(std::string("LoRaDetectorHip: ") + lorahip_strerror(rc) + " (" + lorahip_last_error() + ")");
    }
    ~LoRaDetectorHip(void) { lorahip_detector_destroy(_det); }
    LoRaDetectorHip(const LoRaDetectorHip &) = delete;
    LoRaDetectorHip &operator=(const LoRaDetectorHip &) = delete;

    //! feed simply sets an input sample                                   LoRaDetector.hpp:23
    void feed(const size_t i, const std::complex<Type> &samp)
    {
        // i >= N is undefined behaviour in the reference (a write past _fftInput); here it is an error the caller hears about
        if (lorahip_detector_feed(_det, i, samp.real(), samp.imag()) != LORAHIP_OK)
            throw std::out_of_range("LoRaDetectorHip::feed: sample index outside the window");
    }

    //! calculates argmax(abs(fft(input)))                                 LoRaDetector.hpp:29
    size_t detect(Type &power, Type &powerAvg, Type &fIndex, std::complex<Type> *fftOutput = nullptr)
    {
        size_t index = 0;
        const int rc = lorahip_detector_detect(_det, &index, &power, &powerAvg, &fIndex, reinterpret_cast<float *>(fftOutput));
        if (rc != LORAHIP_OK)
            throw std::runtime_error(std::string("LoRaDetectorHip::detect: ") + lorahip_strerror(rc) + " (" + lorahip_last_error() + ")");
        return index;
    }

private:
    const size_t N;
    lorahip_detector *_det;
};
