"""Python mirror of crates/whacknet/src/lib.rs (the Rust side's `Comm` resource): the 64-byte measurement datagram, the
sender and the gyro listener.  The C++ host layer carries the same classes (include/chalkydri.hpp: whacknet::{WhacknetClient,
Comm, decode_gyro}); this file exists so that the Python parity tests and tools can sit on the same wire."""
import queue
import socket
import struct
import threading

from ._abi import VisionMeasurement

BIND_ADDR = ("0.0.0.0", 0)            # lib.rs:13
REMOTE_ADDR = ("10.45.33.2", 7001)    # lib.rs:14
GYRO_PORT = 7002                      # lib.rs:113


def decode_gyro(buf):
    """One little-endian f64 per datagram (lib.rs:116-123); None for a datagram shorter than 8 bytes."""
    return struct.unpack("<d", bytes(buf[:8]))[0] if len(buf) >= 8 else None


class WhacknetClient:
    """lib.rs:68-89: a UDP socket bound to 0.0.0.0:0 and connected to the roboRIO; send() = the record's 64 raw bytes."""

    def __init__(self, remote=REMOTE_ADDR):
        self._s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        self._s.bind(BIND_ADDR)
        self._s.connect(remote)

    def send(self, m: VisionMeasurement):
        return self._s.send(bytes(m)) == 64

    def close(self):
        self._s.close()


class Comm:
    """lib.rs:99-185: gyro listener thread (latest heading, 0.0 before the first datagram) + publisher thread."""

    def __init__(self, gyro_port=GYRO_PORT, remote=REMOTE_ADDR):
        self._gyro = 0.0                                   # `Some(0f64)` (lib.rs:108)
        self._stop = threading.Event()
        self._gs = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        self._gs.bind(("0.0.0.0", gyro_port))
        self._gs.settimeout(0.1)
        self.gyro_port = self._gs.getsockname()[1]
        self._client = WhacknetClient(remote)
        self._q = queue.Queue()
        self._threads = [threading.Thread(target=self._listen, daemon=True), threading.Thread(target=self._send, daemon=True)]
        for t in self._threads:
            t.start()

    def _listen(self):
        while not self._stop.is_set():
            try:
                data = self._gs.recv(8)
            except OSError:                                # Err(_) => {} (lib.rs:125): timeouts included
                continue
            self._gyro = struct.unpack("<d", data.ljust(8, b"\0")[:8])[0]   # the buffer is zeroed per datagram (lib.rs:128)

    def _send(self):
        while True:
            m = self._q.get()
            if m is None:
                return
            try:
                self._client.send(m)                       # `.ok()`: a failed send is dropped (lib.rs:142)
            except OSError:
                pass

    def publish(self, cam_id, tag_count, ts, pose, std_devs):
        """lib.rs:154-172; pose = (x, y, rot), std_devs = (x, y, rot)."""
        m = VisionMeasurement()
        m.pose_x, m.pose_y, m.pose_rot = pose
        m.std_x, m.std_y, m.std_rot = std_devs
        m.ts, m.camera_id, m.tag_count = ts, cam_id, tag_count
        self._q.put(m)

    def gyro_angle(self):
        return self._gyro                                  # lib.rs:174-179

    def close(self):
        """Drop (lib.rs:180-185): both threads end."""
        self._stop.set()
        self._q.put(None)
        for t in self._threads:
            t.join()
        self._gs.close()
        self._client.close()
