// uninit_vector.h -- a std::vector whose resize() leaves new elements uninitialised: gigabytes of nodes are written once, by
// the builder or by a download, not zeroed first by one thread (assign / resize(n, value) still fill).
#pragma once
#include <memory>
#include <new>
#include <utility>
#include <vector>

namespace chroma_host {
template <class T> struct default_init_allocator : std::allocator<T> {
    template <class U> struct rebind { using other = default_init_allocator<U>; };
    using std::allocator<T>::allocator;
    template <class U> void construct(U *p) noexcept { ::new ((void *)p) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};
template <class T> using uninit_vector = std::vector<T, default_init_allocator<T>>;
}  // namespace chroma_host
