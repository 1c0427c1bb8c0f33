// Shared declarations of libvdbmi: status codes, error text, HIP error plumbing.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "../../include/vdbmi.h"

namespace vdb {

// Error carried across the C boundary as (status, Display text of the matching DBError,
// src/utils/errors.rs:9-33).
struct Error {
    vdb_status code;
    std::string msg;
};
inline Error err_length_mismatch(size_t expected, size_t actual) {       // errors.rs:13
    return {VDB_ERR_VECTOR_LENGTH_MISMATCH,
            "Vector length mismatch: expected " + std::to_string(expected) + ", got " + std::to_string(actual)};
}
inline Error err_invalid_payload(const std::string& s) { return {VDB_ERR_INVALID_PAYLOAD, "Invalid payload: " + s}; }   // errors.rs:28
inline Error err_search(const std::string& s) { return {VDB_ERR_SEARCH, "Search failed: " + s}; }                     // errors.rs:31
inline Error err_not_found(uint64_t id) { return {VDB_ERR_NOT_FOUND, "Point with ID " + std::to_string(id) + " not found"}; }   // errors.rs:10
inline Error err_panic(const std::string& s) { return {VDB_ERR_PANIC, s}; }
inline Error err_arg(const std::string& s) { return {VDB_ERR_ARGUMENT, s}; }
inline Error err_io(const std::string& s) { return {VDB_ERR_IO, "I/O error: " + s}; }                                    // errors.rs:19
inline Error err_device(const std::string& s) { return {VDB_ERR_DEVICE, s}; }

struct Failure : std::exception {
    Error e;
    explicit Failure(Error err) : e(std::move(err)) {}
    const char* what() const noexcept override { return e.msg.c_str(); }
};

#define VDB_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            throw ::vdb::Failure(::vdb::err_device(std::string(#expr) + ": " + hipGetErrorString(_e))); \
    } while (0)

void set_last_error(const std::string& s);

// Counter RNG shared by level assignment and reservoir sampling (see the RNG line of the
// determinism contract in vdbmi.h).
inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t rng_draw(uint64_t seed, uint64_t stream, uint64_t counter) {
    return splitmix64(splitmix64(seed ^ (stream * 0xD1B54A32D192ED03ull)) + counter);
}

}  // namespace vdb
