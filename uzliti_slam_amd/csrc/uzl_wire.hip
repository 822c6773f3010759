// uzl_wire.hip — host side of the wire / disk formats either side of the path:
//   graph_slam_msgs/{Edge,Node,SensorData,Features,Feature}.msg in ROS 1 serialisation <-> the graph objects, as
//   Conversions (graph_slam_common/src/conversions.cpp:43-70,217-322) and SensorData / FeatureData::toMsg / fromMsg
//   (graph_slam_common/src/sensor_data.cpp:40-167) convert them, and the one-message-per-file rosbag 2.0 container of
//   RosbagStorage (graph_slam_common/src/rosbag_storage.cpp:62-209).
// Only message headers, strings and fixed-size fields are handled here (host work: a few dozen fields per message);
// the Feature[] payload is located and handed on as a byte span - the device unpacks / packs it (wire_kernels.hip).
// Built with -ffp-contract=off: the pose <-> quaternion arithmetic rounds after every operation.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/uzl_mi355x.h"

namespace {

// ---------------------------------------------------------------------------------------------- byte streams
struct Reader {
    const uint8_t* b;
    uint64_t n, o = 0;
    bool ok = true;
    Reader(const uint8_t* buf, uint64_t len) : b(buf), n(len) {}
    bool need(uint64_t k)
    {
        if (!ok || k > n - o) { ok = false; return false; }
        return true;
    }
    template <typename T> T get()
    {
        T v{};
        if (need(sizeof(T))) { memcpy(&v, b + o, sizeof(T)); o += sizeof(T); }
        return v;
    }
    void skip(uint64_t k) { if (need(k)) o += k; }
    uzl_span bytes(uint64_t k)
    {
        uzl_span s{nullptr, 0};
        if (need(k)) { s.p = reinterpret_cast<const char*>(b + o); s.n = k; o += k; }
        return s;
    }
    uzl_span str() { const uint32_t l = get<uint32_t>(); return bytes(l); }
    template <typename T> void skip_array() { const uint32_t c = get<uint32_t>(); skip((uint64_t)c * sizeof(T)); }
};

// b == nullptr: counts only (the *_size entry points)
struct Writer {
    uint8_t* b;
    uint64_t cap, o = 0;
    Writer(uint8_t* buf, uint64_t c) : b(buf), cap(c) {}
    void put(const void* p, uint64_t k)
    {
        if (b && k <= cap && o <= cap - k && k) memcpy(b + o, p, k);
        o += k;
    }
    template <typename T> void val(T v) { put(&v, sizeof(T)); }
    void zeros(uint64_t k)
    {
        if (b && k <= cap && o <= cap - k) memset(b + o, 0, k);
        o += k;
    }
    void fill(uint8_t c, uint64_t k)
    {
        if (b && k <= cap && o <= cap - k) memset(b + o, c, k);
        o += k;
    }
    void span(const uzl_span& s) { put(s.p, s.p ? s.n : 0); }
    void str(const uzl_span& s) { val<uint32_t>(s.p ? (uint32_t)s.n : 0u); span(s); }
    bool fits() const { return !b || o <= cap; }
};

// ---------------------------------------------------------------------------------------------- poses
// Eigen::Quaterniond(Matrix3d) [EXT Eigen 3.2, quaternionbase_assign_impl<Other,3,3>], as Conversions::toMsg uses it
// (conversions.cpp:57-70): no normalisation, no sign convention.  q = (x, y, z, w) in wire order.
void quat_from_rotation(const double T[12], double q[4])
{
    const double m00 = T[0], m01 = T[1], m02 = T[2], m10 = T[4], m11 = T[5], m12 = T[6], m20 = T[8], m21 = T[9], m22 = T[10];
    double t = m00 + m11 + m22;
    if (t > 0.) {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m21 - m12) * t;
        q[1] = (m02 - m20) * t;
        q[2] = (m10 - m01) * t;
    } else {
        const double m[3][3] = {{m00, m01, m02}, {m10, m11, m12}, {m20, m21, m22}};
        int i = 0;
        if (m11 > m00) i = 1;
        if (m22 > m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[k][j] - m[j][k]) * t;
        q[j] = (m[j][i] + m[i][j]) * t;
        q[k] = (m[k][i] + m[i][k]) * t;
    }
}

// g2o::internal::fromVectorQT (isometry3d_mappings.cpp:131-136) = Quaterniond(w,x,y,z).toRotationMatrix() [EXT Eigen],
// as Conversions::fromMsg uses it (conversions.cpp:229-240): the quaternion is NOT normalised
void pose_from_wire(const double p[3], const double q[4], double T[12])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    T[0] = 1 - (tyy + tzz); T[1] = txy - twz;       T[2] = txz + twy;        T[3] = p[0];
    T[4] = txy + twz;       T[5] = 1 - (txx + tzz); T[6] = tyz - twx;        T[7] = p[1];
    T[8] = txz - twy;       T[9] = tyz + twx;       T[10] = 1 - (txx + tyy); T[11] = p[2];
}

void put_pose(Writer& w, const double T[12])          // geometry_msgs/Pose: position xyz, orientation xyzw
{
    double q[4];
    quat_from_rotation(T, q);
    w.val(T[3]); w.val(T[7]); w.val(T[11]);
    for (int i = 0; i < 4; i++) w.val(q[i]);
}

void get_pose(Reader& r, double T[12])
{
    double p[3], q[4];
    for (int i = 0; i < 3; i++) p[i] = r.get<double>();
    for (int i = 0; i < 4; i++) q[i] = r.get<double>();
    pose_from_wire(p, q, T);
}

// ---------------------------------------------------------------------------------------------- std / sensor msgs
struct Header { uint32_t seq, sec, nsec; uzl_span frame_id; };
Header get_header(Reader& r)
{
    Header h;
    h.seq = r.get<uint32_t>(); h.sec = r.get<uint32_t>(); h.nsec = r.get<uint32_t>(); h.frame_id = r.str();
    return h;
}
void put_header(Writer& w, uint32_t sec, uint32_t nsec, const uzl_span& frame)
{
    w.val<uint32_t>(0); w.val(sec); w.val(nsec); w.str(frame);
}
void skip_camera_info(Reader& r)      // sensor_msgs/CameraInfo
{
    get_header(r);
    r.skip(8);                        // height, width
    r.str();                          // distortion_model
    r.skip_array<double>();           // D
    r.skip((9 + 9 + 12) * 8);         // K, R, P
    r.skip(8);                        // binning_x, binning_y
    r.skip(17);                       // RegionOfInterest: 4 x uint32 + bool
}
void put_default_camera_info(Writer& w) { w.zeros(16 + 8 + 4 + 4 + 240 + 8 + 17); }
void skip_image(Reader& r)            // sensor_msgs/Image
{
    get_header(r);
    r.skip(8);
    r.str();
    r.skip(1 + 4);
    r.skip_array<uint8_t>();
}
void put_default_image(Writer& w) { w.zeros(16 + 8 + 4 + 1 + 4 + 4); }
void skip_laser_scan(Reader& r)       // sensor_msgs/LaserScan
{
    get_header(r);
    r.skip(7 * 4);
    r.skip_array<float>();
    r.skip_array<float>();
}
void put_default_laser_scan(Writer& w) { w.zeros(16 + 28 + 4 + 4); }

constexpr uint64_t kFeatureFixed = 4 + 4 + 1 + 4 + 4 + 24;       // u, v, is_3d, keypoint_strength, descriptor count, keypoint_position

// graph_slam_msgs/SensorData (SensorData.msg): fields in declaration order
void get_sensor(Reader& r, uzl_wire_sensor* out)
{
    uzl_wire_sensor s;
    memset(&s, 0, sizeof(s));
    const uint64_t start = r.o;
    const Header h = get_header(r);
    s.stamp_sec = h.sec; s.stamp_nsec = h.nsec; s.sensor_frame = h.frame_id;     // SensorData::fromMsg (sensor_data.cpp:52-58)
    s.sensor_type = r.get<int32_t>();
    get_pose(r, s.displacement);
    r.str();                                                                     // sensor_frame (fromMsg reads the header's)
    // Features features
    get_header(r);
    s.descriptor_type = r.get<int32_t>();
    const uint32_t nf = r.get<uint32_t>();
    const uint64_t rec0 = r.o;
    s.uniform = 1;
    for (uint32_t i = 0; i < nf && r.ok; i++) {
        r.skip(13);
        const uint32_t d = r.get<uint32_t>();
        if (i == 0) s.desc_len = (int32_t)d;
        else if ((int32_t)d != s.desc_len) s.uniform = 0;
        r.skip((uint64_t)d * 4 + 24);
    }
    s.n_features = (int32_t)nf;
    if (r.ok) { s.records.p = reinterpret_cast<const char*>(r.b + rec0); s.records.n = r.o - rec0; }
    const uint64_t cam0 = r.o;
    skip_camera_info(r);
    if (r.ok) { s.camera_info.p = reinterpret_cast<const char*>(r.b + cam0); s.camera_info.n = r.o - cam0; }
    skip_image(r); skip_image(r);                                                // DepthImage: depth, color
    r.skip_array<float>();                                                       // gist_descriptor
    skip_laser_scan(r);
    r.skip(24);                                                                  // scan_center
    if (r.ok) { s.raw.p = reinterpret_cast<const char*>(r.b + start); s.raw.n = r.o - start; }
    if (out) *out = s;
}

void put_sensor(Writer& w, const uzl_wire_sensor& s)
{
    if (s.raw.p) { w.span(s.raw); return; }
    // SensorData::toMsg (sensor_data.cpp:40-49) + FeatureData::toMsg (:78-121)
    put_header(w, s.stamp_sec, s.stamp_nsec, s.sensor_frame);
    w.val<int32_t>(s.sensor_type);
    put_pose(w, s.displacement);
    w.str(s.sensor_frame);
    put_header(w, s.stamp_sec, s.stamp_nsec, s.sensor_frame);                    // features.header (:82-83)
    w.val<int32_t>(s.descriptor_type);
    w.val<uint32_t>((uint32_t)s.n_features);
    w.span(s.records);
    if (s.camera_info.p) w.span(s.camera_info); else put_default_camera_info(w);
    put_default_image(w); put_default_image(w);
    w.val<uint32_t>(0);
    put_default_laser_scan(w);
    w.zeros(24);
}

void put_edge(Writer& w, const uzl_wire_edge& e)       // Conversions::toMsg(SlamEdge) (conversions.cpp:255-274), Edge.msg order
{
    w.str(e.id);
    w.val<uint8_t>((uint8_t)e.type);
    w.str(e.id_from); w.str(e.id_to);
    put_pose(w, e.displacement_from); put_pose(w, e.displacement_to);
    put_pose(w, e.transform);
    for (int i = 0; i < 36; i++) w.val(e.information[i]);                        // toMsg(pose, Sigma) (:43-55)
    w.val(e.error); w.val(e.age); w.val(e.matching_score);
    w.str(e.sensor_from); w.str(e.sensor_to);
    w.val<uint8_t>(e.valid ? 1 : 0);
    w.val<int32_t>(e.diff_time_sec); w.val<int32_t>(e.diff_time_nsec);
}

void put_node(Writer& w, const uzl_wire_node& n, const int64_t* stamps_ns, const uzl_span* edge_ids, const uzl_wire_sensor* sensors)
{
    // Conversions::toMsg(SlamNode) (conversions.cpp:299-322), Node.msg order
    w.val<uint32_t>((uint32_t)n.n_stamps);
    for (int32_t i = 0; i < n.n_stamps; i++) {
        const int64_t t = stamps_ns ? stamps_ns[i] : 0;
        w.val<uint32_t>((uint32_t)(t / 1000000000)); w.val<uint32_t>((uint32_t)(t % 1000000000));
    }
    w.str(n.id);
    put_pose(w, n.pose); put_pose(w, n.odom_pose);
    put_header(w, 0, 0, uzl_span{nullptr, 0});                                   // sensor_data.header stays default
    w.val<uint32_t>((uint32_t)n.n_sensors);
    for (int32_t i = 0; i < n.n_sensors; i++) put_sensor(w, sensors[i]);
    w.val<uint32_t>((uint32_t)n.n_edge_ids);
    for (int32_t i = 0; i < n.n_edge_ids; i++) w.str(edge_ids[i]);
    w.val<uint8_t>(n.fixed ? 1 : 0);
    w.val(n.uncertainty);
}

// ---------------------------------------------------------------------------------------------- rosbag 2.0
constexpr char kBagMagic[] = "#ROSBAG V2.0\n";
constexpr uint64_t kBagMagicLen = 13;
constexpr uint64_t kBagFileHeaderLength = 4096;        // rosbag FILE_HEADER_LENGTH: writeFileHeaderRecord sets data_len = 4096 - header_len,
                                                       // so the whole record is 4 + header_len + 4 + (4096 - header_len) = 4104 bytes

struct Field { uzl_span name, value; };
// one record header: fields "name=value", each with a u32 length
bool parse_fields(const uint8_t* p, uint64_t len, std::vector<Field>& out)
{
    out.clear();
    Reader r(p, len);
    while (r.ok && r.o < r.n) {
        const uint32_t fl = r.get<uint32_t>();
        const uzl_span f = r.bytes(fl);
        if (!r.ok || f.n == 0) return false;
        const char* eq = static_cast<const char*>(memchr(f.p, '=', f.n));
        if (!eq) return false;
        Field fd;
        fd.name = uzl_span{f.p, (uint64_t)(eq - f.p)};
        fd.value = uzl_span{eq + 1, f.n - (uint64_t)(eq - f.p) - 1};
        out.push_back(fd);
    }
    return r.ok;
}
const Field* find(const std::vector<Field>& fs, const char* name)
{
    const size_t l = strlen(name);
    for (const Field& f : fs) if (f.name.n == l && memcmp(f.name.p, name, l) == 0) return &f;
    return nullptr;
}
bool field_u32(const std::vector<Field>& fs, const char* name, uint32_t* v)
{
    const Field* f = find(fs, name);
    if (!f || f->value.n != 4) return false;
    memcpy(v, f->value.p, 4);
    return true;
}
bool field_u8(const std::vector<Field>& fs, const char* name, uint8_t* v)
{
    const Field* f = find(fs, name);
    if (!f || f->value.n != 1) return false;
    *v = (uint8_t)f->value.p[0];
    return true;
}

struct Conn { uint32_t id; uzl_span topic, type, md5, def; };

struct BagScan {
    std::vector<Conn> conns;
    int32_t cap = 0, found = 0;
    uzl_bag_msg* out = nullptr;
    int status = UZL_OK;
};

// Walks the records in [p, p+len); pass 0 collects connections, pass 1 emits messages.
void scan_records(const uint8_t* p, uint64_t len, int pass, bool in_chunk, BagScan& S)
{
    Reader r(p, len);
    std::vector<Field> fs;
    while (r.o < r.n) {
        const uint32_t hl = r.get<uint32_t>();
        const uzl_span hd = r.bytes(hl);
        const uint32_t dl = r.get<uint32_t>();
        const uzl_span data = r.bytes(dl);
        if (!r.ok) { S.status = UZL_ERR_TRUNCATED; return; }
        if (!parse_fields(reinterpret_cast<const uint8_t*>(hd.p), hd.n, fs)) { S.status = UZL_ERR_BAD_ARG; return; }
        uint8_t op = 0;
        if (!field_u8(fs, "op", &op)) { S.status = UZL_ERR_BAD_ARG; return; }
        if (op == 0x05 && !in_chunk) {                                           // chunk
            const Field* c = find(fs, "compression");
            if (!c || c->value.n != 4 || memcmp(c->value.p, "none", 4) != 0) { S.status = UZL_ERR_UNSUPPORTED; return; }
            scan_records(reinterpret_cast<const uint8_t*>(data.p), data.n, pass, true, S);
            if (S.status != UZL_OK) return;
        } else if (op == 0x07 && pass == 0) {                                    // connection
            Conn c;
            memset(&c, 0, sizeof(c));
            if (!field_u32(fs, "conn", &c.id)) { S.status = UZL_ERR_BAD_ARG; return; }
            bool seen = false;
            for (const Conn& k : S.conns) seen = seen || k.id == c.id;
            if (seen) continue;
            if (const Field* t = find(fs, "topic")) c.topic = t->value;
            std::vector<Field> cf;
            if (!parse_fields(reinterpret_cast<const uint8_t*>(data.p), data.n, cf)) { S.status = UZL_ERR_BAD_ARG; return; }
            if (const Field* f = find(cf, "type")) c.type = f->value;
            if (const Field* f = find(cf, "md5sum")) c.md5 = f->value;
            if (const Field* f = find(cf, "message_definition")) c.def = f->value;
            S.conns.push_back(c);
        } else if (op == 0x02 && pass == 1) {                                    // message data
            uint32_t id = 0;
            const Field* t = find(fs, "time");
            if (!field_u32(fs, "conn", &id) || !t || t->value.n != 8) { S.status = UZL_ERR_BAD_ARG; return; }
            const Conn* c = nullptr;
            for (const Conn& k : S.conns) if (k.id == id) c = &k;
            if (!c) { S.status = UZL_ERR_NOT_FOUND; return; }
            if (S.found < S.cap) {
                uzl_bag_msg& m = S.out[S.found];
                m.topic = c->topic; m.datatype = c->type; m.md5sum = c->md5; m.definition = c->def;
                m.data = data;
                memcpy(&m.time_sec, t->value.p, 4); memcpy(&m.time_nsec, t->value.p + 4, 4);
            }
            S.found++;
        }
    }
}

// record header writer: fields in name order, as rosbag's std::map<string,string> headers come out
struct FieldOut { const char* name; const void* p; uint64_t n; };
uint64_t fields_len(const FieldOut* f, int n)
{
    uint64_t l = 0;
    for (int i = 0; i < n; i++) l += 4 + strlen(f[i].name) + 1 + f[i].n;
    return l;
}
void put_fields(Writer& w, const FieldOut* f, int n)
{
    for (int i = 0; i < n; i++) {
        const uint64_t nl = strlen(f[i].name);
        w.val<uint32_t>((uint32_t)(nl + 1 + f[i].n));
        w.put(f[i].name, nl);
        w.val<uint8_t>('=');
        w.put(f[i].p, f[i].n);
    }
}
void put_record_header(Writer& w, const FieldOut* f, int n)
{
    w.val<uint32_t>((uint32_t)fields_len(f, n));
    put_fields(w, f, n);
}

void put_bag(Writer& w, const uzl_bag_msg& m)
{
    const uint8_t op_msg = 0x02, op_hdr = 0x03, op_idx = 0x04, op_chunk = 0x05, op_info = 0x06, op_conn = 0x07;
    const uint32_t conn = 0, one = 1, ver = 1;
    const uint32_t time[2] = {m.time_sec, m.time_nsec};
    const uzl_span none{nullptr, 0};
    const uzl_span& def = m.definition.p ? m.definition : none;
    // connection record (header: conn, op, topic; data: md5sum, message_definition, type)
    const FieldOut conn_hdr[3] = {{"conn", &conn, 4}, {"op", &op_conn, 1}, {"topic", m.topic.p, m.topic.n}};
    const FieldOut conn_dat[3] = {{"md5sum", m.md5sum.p, m.md5sum.n}, {"message_definition", def.p, def.n}, {"type", m.datatype.p, m.datatype.n}};
    const uint64_t conn_rec = 4 + fields_len(conn_hdr, 3) + 4 + fields_len(conn_dat, 3);
    const FieldOut msg_hdr[3] = {{"conn", &conn, 4}, {"op", &op_msg, 1}, {"time", time, 8}};
    const uint64_t msg_rec = 4 + fields_len(msg_hdr, 3) + 4 + m.data.n;
    const uint32_t chunk_size = (uint32_t)(conn_rec + msg_rec);
    const FieldOut chunk_hdr[3] = {{"compression", "none", 4}, {"op", &op_chunk, 1}, {"size", &chunk_size, 4}};
    const uint64_t chunk_pos = kBagMagicLen + 4 + 4 + kBagFileHeaderLength;   // 4117
    const uint64_t chunk_rec = 4 + fields_len(chunk_hdr, 3) + 4 + chunk_size;
    const FieldOut idx_hdr[4] = {{"conn", &conn, 4}, {"count", &one, 4}, {"op", &op_idx, 1}, {"ver", &ver, 4}};
    const uint64_t idx_rec = 4 + fields_len(idx_hdr, 4) + 4 + 12;
    const uint64_t index_pos = chunk_pos + chunk_rec + idx_rec;
    // file header record: header + space padding = 4096 bytes (rosbag::Bag::writeFileHeaderRecord)
    w.put(kBagMagic, kBagMagicLen);
    const FieldOut bag_hdr[4] = {{"chunk_count", &one, 4}, {"conn_count", &one, 4}, {"index_pos", &index_pos, 8}, {"op", &op_hdr, 1}};
    put_record_header(w, bag_hdr, 4);
    const uint64_t pad = kBagFileHeaderLength - fields_len(bag_hdr, 4);
    w.val<uint32_t>((uint32_t)pad);
    w.fill(' ', pad);
    // chunk
    put_record_header(w, chunk_hdr, 3);
    w.val<uint32_t>(chunk_size);
    put_record_header(w, conn_hdr, 3);
    w.val<uint32_t>((uint32_t)fields_len(conn_dat, 3));
    put_fields(w, conn_dat, 3);
    put_record_header(w, msg_hdr, 3);
    w.val<uint32_t>((uint32_t)m.data.n);
    w.span(m.data);
    // index data of the chunk: (time, offset of the message record inside the chunk)
    put_record_header(w, idx_hdr, 4);
    w.val<uint32_t>(12);
    w.val(time[0]); w.val(time[1]); w.val<uint32_t>((uint32_t)conn_rec);
    // index section: connection, chunk info
    put_record_header(w, conn_hdr, 3);
    w.val<uint32_t>((uint32_t)fields_len(conn_dat, 3));
    put_fields(w, conn_dat, 3);
    const FieldOut info_hdr[6] = {{"chunk_pos", &chunk_pos, 8}, {"count", &one, 4}, {"end_time", time, 8}, {"op", &op_info, 1},
                                  {"start_time", time, 8}, {"ver", &ver, 4}};
    put_record_header(w, info_hdr, 6);
    w.val<uint32_t>(8);
    w.val(conn); w.val(one);
}

}  // namespace

extern "C" {

uint64_t uzl_wire_edge_size(const uzl_wire_edge* e)
{
    if (!e) return 0;
    Writer w(nullptr, 0);
    put_edge(w, *e);
    return w.o;
}

int uzl_wire_edge_encode(const uzl_wire_edge* e, uint8_t* buf, uint64_t cap, uint64_t* written)
{
    if (!e || !buf) return UZL_ERR_BAD_ARG;
    Writer w(buf, cap);
    put_edge(w, *e);
    if (written) *written = w.o;
    return w.fits() ? UZL_OK : UZL_ERR_TRUNCATED;
}

int uzl_wire_edge_decode(const uint8_t* buf, uint64_t len, uzl_wire_edge* out, uint64_t* consumed)
{
    if (!buf || !out) return UZL_ERR_BAD_ARG;
    Reader r(buf, len);
    uzl_wire_edge e;
    memset(&e, 0, sizeof(e));
    // Conversions::fromMsg(Edge) (conversions.cpp:242-253)
    e.id = r.str();
    e.type = r.get<uint8_t>();
    e.id_from = r.str(); e.id_to = r.str();
    get_pose(r, e.displacement_from); get_pose(r, e.displacement_to);
    get_pose(r, e.transform);
    for (int i = 0; i < 36; i++) e.information[i] = r.get<double>();
    e.error = r.get<double>(); e.age = r.get<double>(); e.matching_score = r.get<double>();
    e.sensor_from = r.str(); e.sensor_to = r.str();
    e.valid = r.get<uint8_t>() != 0;
    e.diff_time_sec = r.get<int32_t>(); e.diff_time_nsec = r.get<int32_t>();
    if (!r.ok) return UZL_ERR_TRUNCATED;
    *out = e;
    if (consumed) *consumed = r.o;
    return UZL_OK;
}

// SlamGraph::toMetaData (slam_graph.cpp:592-619) serialised in GraphMeta.msg field order
static void put_meta(Writer& w, const uzl_wire_meta& m, const uzl_wire_sensor_transform* st, const uzl_wire_sensor_transform* sti)
{
    put_header(w, m.stamp_sec, m.stamp_nsec, m.frame_id);
    w.str(m.name);
    put_pose(w, m.map_transform);
    const int32_t n0 = st ? std::max(m.n_sensor_transforms, 0) : 0, n1 = sti ? std::max(m.n_sensor_transforms_initial, 0) : 0;
    w.val<uint32_t>((uint32_t)n0);
    for (int32_t i = 0; i < n0; i++) { w.str(st[i].sensor_name); put_pose(w, st[i].transform); }
    w.val<uint32_t>((uint32_t)n1);
    for (int32_t i = 0; i < n1; i++) { w.str(sti[i].sensor_name); put_pose(w, sti[i].transform); }
    for (int i = 0; i < 6; i++) w.val(m.odometry_parameters[i]);              // Conversions::toMsg(Vector6d) (conversions.cpp:338-343)
}

uint64_t uzl_wire_meta_size(const uzl_wire_meta* m, const uzl_wire_sensor_transform* st, const uzl_wire_sensor_transform* sti)
{
    if (!m) return 0;
    Writer w(nullptr, 0);
    put_meta(w, *m, st, sti);
    return w.o;
}

int uzl_wire_meta_encode(const uzl_wire_meta* m, const uzl_wire_sensor_transform* st, const uzl_wire_sensor_transform* sti, uint8_t* buf,
                         uint64_t cap, uint64_t* written)
{
    if (!m || !buf || (m->n_sensor_transforms > 0 && !st) || (m->n_sensor_transforms_initial > 0 && !sti) || m->n_sensor_transforms < 0 ||
        m->n_sensor_transforms_initial < 0) return UZL_ERR_BAD_ARG;
    Writer w(buf, cap);
    put_meta(w, *m, st, sti);
    if (written) *written = w.o;
    return w.fits() ? UZL_OK : UZL_ERR_TRUNCATED;
}

int uzl_wire_meta_decode(const uint8_t* buf, uint64_t len, uzl_wire_meta* out, int32_t cap, uzl_wire_sensor_transform* st, int32_t cap_initial,
                         uzl_wire_sensor_transform* sti, uint64_t* consumed)
{
    if (!buf || !out) return UZL_ERR_BAD_ARG;
    Reader r(buf, len);
    uzl_wire_meta m;
    memset(&m, 0, sizeof(m));
    // SlamGraph::updateMetaData (slam_graph.cpp:621-633)
    const Header h = get_header(r);
    m.stamp_sec = h.sec; m.stamp_nsec = h.nsec; m.frame_id = h.frame_id;
    m.name = r.str();
    get_pose(r, m.map_transform);
    for (int which = 0; which < 2 && r.ok; which++) {
        const uint32_t c = r.get<uint32_t>();
        uzl_wire_sensor_transform* dst = which ? sti : st;
        const int32_t room = which ? cap_initial : cap;
        // (a count the rest of the message cannot hold: every entry takes at least 4 + 56 bytes)
        if (r.ok && (uint64_t)c > (r.n - r.o) / 60) { r.ok = false; break; }
        for (uint32_t i = 0; i < c && r.ok; i++) {
            uzl_wire_sensor_transform e;
            e.sensor_name = r.str();
            get_pose(r, e.transform);
            if (r.ok && dst && (int64_t)i < room) dst[i] = e;
        }
        (which ? m.n_sensor_transforms_initial : m.n_sensor_transforms) = (int32_t)c;
    }
    for (int i = 0; i < 6; i++) m.odometry_parameters[i] = r.get<double>();
    if (!r.ok) return UZL_ERR_TRUNCATED;
    *out = m;
    if (consumed) *consumed = r.o;
    return UZL_OK;
}

int uzl_wire_node_decode(const uint8_t* buf, uint64_t len, uzl_wire_node* out, int32_t stamp_cap, int64_t* stamps_ns,
                         int32_t edge_cap, uzl_span* edge_ids, int32_t sensor_cap, uzl_wire_sensor* sensors, uint64_t* consumed)
{
    if (!buf || !out) return UZL_ERR_BAD_ARG;
    Reader r(buf, len);
    uzl_wire_node n;
    memset(&n, 0, sizeof(n));
    // Conversions::fromMsg(Node) (conversions.cpp:276-297)
    const uint32_t ns = r.get<uint32_t>();
    for (uint32_t i = 0; i < ns && r.ok; i++) {
        const uint32_t sec = r.get<uint32_t>(), nsec = r.get<uint32_t>();
        if (stamps_ns && (int64_t)i < stamp_cap) stamps_ns[i] = (int64_t)sec * 1000000000 + nsec;
    }
    n.n_stamps = (int32_t)ns;
    n.id = r.str();
    get_pose(r, n.pose); get_pose(r, n.odom_pose);
    get_header(r);                                                               // SensorDataArray.header
    const uint32_t nd = r.get<uint32_t>();
    for (uint32_t i = 0; i < nd && r.ok; i++) get_sensor(r, (sensors && (int64_t)i < sensor_cap) ? sensors + i : nullptr);
    n.n_sensors = (int32_t)nd;
    const uint32_t ne = r.get<uint32_t>();
    for (uint32_t i = 0; i < ne && r.ok; i++) {
        const uzl_span s = r.str();
        if (edge_ids && (int64_t)i < edge_cap) edge_ids[i] = s;
    }
    n.n_edge_ids = (int32_t)ne;
    n.fixed = r.get<uint8_t>() != 0;
    n.uncertainty = r.get<double>();
    if (!r.ok) return UZL_ERR_TRUNCATED;
    *out = n;
    if (consumed) *consumed = r.o;
    return UZL_OK;
}

uint64_t uzl_wire_node_size(const uzl_wire_node* n, const uzl_span* edge_ids, const uzl_wire_sensor* sensors)
{
    if (!n || (n->n_edge_ids > 0 && !edge_ids) || (n->n_sensors > 0 && !sensors)) return 0;
    Writer w(nullptr, 0);
    put_node(w, *n, nullptr, edge_ids, sensors);
    return w.o;
}

int uzl_wire_node_encode(const uzl_wire_node* n, const int64_t* stamps_ns, const uzl_span* edge_ids, const uzl_wire_sensor* sensors,
                         uint8_t* buf, uint64_t cap, uint64_t* written)
{
    if (!n || !buf || n->n_stamps < 0 || n->n_edge_ids < 0 || n->n_sensors < 0) return UZL_ERR_BAD_ARG;
    if ((n->n_stamps > 0 && !stamps_ns) || (n->n_edge_ids > 0 && !edge_ids) || (n->n_sensors > 0 && !sensors)) return UZL_ERR_BAD_ARG;
    for (int32_t i = 0; i < n->n_sensors; i++) {
        const uzl_wire_sensor& s = sensors[i];
        if (!s.raw.p && s.records.n != uzl_wire_features_size(s.n_features, s.desc_len)) return UZL_ERR_BAD_ARG;
    }
    Writer w(buf, cap);
    put_node(w, *n, stamps_ns, edge_ids, sensors);
    if (written) *written = w.o;
    return w.fits() ? UZL_OK : UZL_ERR_TRUNCATED;
}

uint64_t uzl_wire_features_size(int32_t n, int32_t desc_len)
{
    if (n <= 0 || desc_len < 0) return 0;
    return (uint64_t)n * (kFeatureFixed + 4ull * (uint64_t)desc_len);
}

int uzl_bag_read(const uint8_t* file, uint64_t len, int32_t cap, uzl_bag_msg* msgs, int32_t* n_msgs)
{
    if (!file || !n_msgs || cap < 0 || (cap > 0 && !msgs)) return UZL_ERR_BAD_ARG;
    if (len < kBagMagicLen || memcmp(file, kBagMagic, kBagMagicLen) != 0) return UZL_ERR_BAD_ARG;
    BagScan S;
    S.cap = cap; S.out = msgs;
    for (int pass = 0; pass < 2; pass++) {
        scan_records(file + kBagMagicLen, len - kBagMagicLen, pass, false, S);
        if (S.status != UZL_OK) return S.status;
    }
    *n_msgs = S.found;
    return UZL_OK;
}

uint64_t uzl_bag_single_size(const uzl_bag_msg* m)
{
    if (!m) return 0;
    Writer w(nullptr, 0);
    put_bag(w, *m);
    return w.o;
}

int uzl_bag_write_single(const uzl_bag_msg* m, uint8_t* out, uint64_t cap, uint64_t* written)
{
    if (!m || !out || !m->topic.p || !m->datatype.p || !m->md5sum.p || (m->data.n && !m->data.p)) return UZL_ERR_BAD_ARG;
    Writer w(out, cap);
    put_bag(w, *m);
    if (written) *written = w.o;
    return w.fits() ? UZL_OK : UZL_ERR_TRUNCATED;
}

}  // extern "C"
