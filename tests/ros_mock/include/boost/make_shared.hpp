// MOCK (tests/ros_mock): boost::shared_ptr / make_shared / shared_array as aliases of the standard ones -- enough for the
// adapter's sources to compile in a container without Boost.  Not Boost.
#pragma once
#include <memory>
namespace boost {
template <class T> using shared_ptr = std::shared_ptr<T>;
template <class T, class... A> shared_ptr<T> make_shared(A&&... a) { return std::make_shared<T>(std::forward<A>(a)...); }
template <class T> class shared_array {
 public:
  shared_array() = default;
  explicit shared_array(T* p) : p_(p, std::default_delete<T[]>()) {}
  T* get() const { return p_.get(); }
 private:
  std::shared_ptr<T> p_;
};
}  // namespace boost
