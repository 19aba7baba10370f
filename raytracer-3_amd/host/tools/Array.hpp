// tools/Array.hpp — Tools::Array<T>, the container type in the Renderer boundary signature
// (reference: src/lib/tools/Array.hpp:28-117).  Only the surface the boundary and its callers use is provided,
// on top of std::vector; the reference's malloc-based internals are out of scope.
// Kept gotchas: Array(n) reserves capacity WITHOUT changing size() (Array.cpp:36-43); clear() releases storage.
#ifndef RT3_HOST_TOOLS_ARRAY_HPP
#define RT3_HOST_TOOLS_ARRAY_HPP
#include <cstddef>
#include <initializer_list>
#include <stdexcept>
#include <vector>

namespace Tools {
template <class T>
class Array {
    std::vector<T> items;

public:
    Array() = default;
    explicit Array(size_t initial_capacity) { items.reserve(initial_capacity); }
    Array(const std::initializer_list<T>& list) : items(list) {}
    Array(const T* list, size_t n) : items(list, list + n) {}
    explicit Array(const std::vector<T>& list) : items(list) {}

    Array& operator+=(const Array& other) { items.insert(items.end(), other.items.begin(), other.items.end()); return *this; }
    void push_back(const T& v) { items.push_back(v); }
    void push_back(T&& v) { items.push_back(std::move(v)); }
    void pop_back() { items.pop_back(); }
    void clear() { std::vector<T>().swap(items); }
    void reserve(size_t n) { if (n < items.size()) items.resize(n); items.reserve(n); }   // shrinking truncates (Array.cpp:262-289)
    void resize(size_t n) { items.resize(n); }

    T& operator[](size_t i) { return items[i]; }
    const T& operator[](size_t i) const { return items[i]; }
    T& at(size_t i) { if (i >= items.size()) throw std::out_of_range("Array::at"); return items[i]; }
    const T& at(size_t i) const { if (i >= items.size()) throw std::out_of_range("Array::at"); return items[i]; }
    T* wdata(size_t new_size = (size_t)-1) { if (new_size != (size_t)-1) items.resize(new_size); return items.data(); }
    const T* rdata() const { return items.data(); }
    bool empty() const { return items.empty(); }
    size_t size() const { return items.size(); }
    size_t capacity() const { return items.capacity(); }
};
}  // namespace Tools
#endif
