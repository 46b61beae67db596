// roscpp stand-in for the node shims (see README.md)
#pragma once
#include <cstdio>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <typeinfo>
#include <vector>

#include <std_msgs/Header.h>

#define ROS_ERROR(...) do { std::fprintf(stderr, "[ERROR] "); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_WARN(...) do { std::fprintf(stderr, "[WARN] "); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_INFO(...) do { std::fprintf(stderr, "[INFO] "); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)

namespace ros {
namespace stub {
struct Param { int kind = 0; bool b = false; double d = 0; std::string s; };   // kind 1 bool, 2 double, 3 string
inline std::map<std::string, Param>& params() { static std::map<std::string, Param> p; return p; }
inline void set(const std::string& k, bool v) { Param p; p.kind = 1; p.b = v; params()[k] = p; }
inline void set(const std::string& k, double v) { Param p; p.kind = 2; p.d = v; params()[k] = p; }
inline void set(const std::string& k, const char* v) { Param p; p.kind = 3; p.s = v; params()[k] = p; }
inline bool get(const std::string& k, bool& v) { auto it = params().find(k); if (it == params().end() || it->second.kind != 1) return false; v = it->second.b; return true; }
inline bool get(const std::string& k, double& v) { auto it = params().find(k); if (it == params().end() || it->second.kind != 2) return false; v = it->second.d; return true; }
inline bool get(const std::string& k, std::string& v) { auto it = params().find(k); if (it == params().end() || it->second.kind != 3) return false; v = it->second.s; return true; }
struct Sent { std::shared_ptr<void> msg; const std::type_info* type = nullptr; int count = 0; };
inline std::map<std::string, Sent>& sent() { static std::map<std::string, Sent> s; return s; }
template <class M> const M* last(const std::string& topic) {
    auto it = sent().find(topic);
    if (it == sent().end() || *it->second.type != typeid(M)) return nullptr;
    return static_cast<const M*>(it->second.msg.get());
}
inline int count(const std::string& topic) { auto it = sent().find(topic); return it == sent().end() ? 0 : it->second.count; }
// what a node's main() registered: advertised topics (with the message type), subscriptions (with the queue size), services
struct Endpoint { std::string name; const std::type_info* type = nullptr; uint32_t queue = 0; };
inline std::vector<Endpoint>& advertised() { static std::vector<Endpoint> v; return v; }
inline std::vector<Endpoint>& subscribed() { static std::vector<Endpoint> v; return v; }
inline std::vector<std::string>& services() { static std::vector<std::string> v; return v; }
template <class M> bool advertises(const std::string& topic) {
    for (const Endpoint& e : advertised()) if (e.name == topic && *e.type == typeid(M)) return true;
    return false;
}
template <class M> bool subscribes(const std::string& topic, uint32_t queue) {
    for (const Endpoint& e : subscribed()) if (e.name == topic && *e.type == typeid(M) && e.queue == queue) return true;
    return false;
}
}  // namespace stub

inline void init(int&, char**, const std::string&) {}
inline void spin() {}
inline bool ok() { return true; }

class Publisher {
  public:
    Publisher() {}
    explicit Publisher(const std::string& t) : topic_(t) {}
    template <class M> void publish(const M& m) const {
        stub::Sent& s = stub::sent()[topic_];
        s.msg = std::make_shared<M>(m);
        s.type = &typeid(M);
        s.count += 1;
    }
    const std::string& getTopic() const { return topic_; }
  private:
    std::string topic_;
};
class Subscriber { public: ~Subscriber() {} };
class ServiceServer { public: ~ServiceServer() {} };

class NodeHandle {
  public:
    explicit NodeHandle(const std::string& ns = "") : ns_(ns) {}
    template <class T> bool param(const std::string& name, T& v, const T& def) const { if (stub::get(name, v)) return true; v = def; return false; }
    template <class T> bool getParam(const std::string& name, T& v) const { return stub::get(name, v); }
    template <class M> Subscriber subscribe(const std::string& t, uint32_t q, void (*)(const std::shared_ptr<const M>&)) { stub::subscribed().push_back({t, &typeid(M), q}); return Subscriber(); }
    template <class M> Subscriber subscribe(const std::string& t, uint32_t q, void (*)(const M&)) { stub::subscribed().push_back({t, &typeid(M), q}); return Subscriber(); }
    template <class M> Publisher advertise(const std::string& topic, uint32_t q) { stub::advertised().push_back({topic, &typeid(M), q}); return Publisher(topic); }
    template <class Req, class Res> ServiceServer advertiseService(const std::string& name, bool (*)(Req&, Res&)) { stub::services().push_back(name); return ServiceServer(); }
  private:
    std::string ns_;
};
}  // namespace ros
